/* libhortihip.so -- C ABI of the MI355X-native HortiMapping hot path.
 *
 * The reference (PRBonn/HortiMapping @ 2024_08_07) is pure Python; its operator interface for this path is
 *     Optimizer(cfg, decoder, mesher, vis).shape_pose_joint_opt(latent, T_ow, render_data, points_w, cube_radius,
 *                                                              cur_color, pose_known) -> (latent, T_ow, iter_count)
 *     Optimizer.shape_opt_deepsdf(latent, T_ow, points_w, cur_color)                 -> (latent, T_ow, iter_count)
 * (wild_completion/optimizer.py:17,28,302,306,429) plus the functional helpers decode_sdf / get_batch_sdf_jacobian
 * (wild_completion/utils.py:144,175) and compute_sdf_loss / compute_render_loss (wild_completion/loss.py:8,219).
 * Each entry point below names the reference interface it replaces.  Conventions:
 *   - every function returns 0 on success, < 0 on error (never throws); hm_last_error() describes the failure;
 *   - all `d_*` pointers are device (HBM) pointers owned by the caller; the library never frees caller memory;
 *   - work is enqueued on the caller's hipStream_t (passed as void*); nothing synchronises the host unless stated
 *     (stated exceptions: hm_optimize_batch's host pacing when early exits are possible -- see there and
 *     hm_workspace_set_host_pacing --, hm_workspace_counters_read, the *_create functions);
 *   - matrices are row-major fp32; index arrays are int32.
 */
#ifndef HORTIMAPPING_AMD_H
#define HORTIMAPPING_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hm_decoder_s* hm_decoder_t;
typedef struct hm_workspace_s* hm_workspace_t;

/* per-instance status bits written by hm_optimize_batch */
#define HM_STATUS_CONV_G 1        /* optimizer.py:276  '**** Convergence in gradient ****'          */
#define HM_STATUS_CONV_C 2        /* optimizer.py:280  '**** Convergence in Shape Latent Code ****' */
#define HM_STATUS_CONV_P 4        /* optimizer.py:285  '**** Convergence in Pose Parameters ****'   */
#define HM_STATUS_MAX_ITER 8      /* optimizer.py:289  maximum iteration number                      */
#define HM_STATUS_INVALID 16      /* optimizer.py:139-141 'This submap is not valid' (no depth residual left) */
#define HM_STATUS_SOLVE_FAILED 32 /* normal matrix not positive definite / non-finite step             */
#define HM_STATUS_FRAME_SKIPPED 64 /* optimizer.py:130-132 'This frame is not valid': in some iteration a frame had
                                      fewer than min_valid_sample ball-valid samples and was left out (informational) */
#define HM_STATUS_LIMIT 128       /* the instance does not fit the workspace (n_points > max_points, n_frames >
                                      max_frames, n_fg + n_bg > max_rays, or more Jacobian samples than
                                      max_grad_samples): it is NOT optimised (iter_count 0, state untouched) or, for
                                      the Jacobian-sample cap, stopped at the iteration that overflowed.  The reference
                                      has no capacities; this bit is how a truncation is reported instead of hidden */

const char* hm_last_error(void);

/* ---- decoder handle: replaces config_decoder (deepsdf/deep_sdf/workspace.py:203-225) + Decoder.__init__
 * (deepsdf/networks/deep_sdf_decoder.py:11-72).  W[l], bias[l] (l = 0..8) are HOST pointers to the folded
 * (weight-norm applied) row-major fp32 matrices: W0 (512, L+3), W1..2 (512,512), W3 (509-L, 512), W4..7 (512,512),
 * W8 (1,512).  Only the shipped architecture (8 x 512, latent_in=[4]) with L a multiple of 32, 32 <= L <= 256; every
 * other layer table goes through hm_decoder_create_arch below. */
int hm_decoder_create(int latent_dim, const float* const* W, const float* const* bias, hm_decoder_t* out);
int hm_decoder_destroy(hm_decoder_t dec);
int hm_decoder_latent_dim(hm_decoder_t dec);

/* ---- decoder handle for ANY layer table the reference's `Decoder` class can build
 * (deepsdf/networks/deep_sdf_decoder.py:11-72: `dims`, `latent_in`, `xyz_in_all`, `norm_layers` with or without
 * `weight_norm`, `use_tanh`; dropout is the identity in eval mode, which is all the optimiser uses).  Layer l
 * (0 <= l < n_lin) is Linear(in_dim[l], out_dim[l]); its input is the previous layer's output, with the network input
 * [z | xyz] appended when cat[l] == 1 (l in latent_in, :87-88) or xyz appended when cat[l] == 2 (xyz_in_all, :89-90);
 * hidden layers are followed by LayerNorm when layer_norm[l] (:96-101; eps 1e-5, affine) and ReLU; the last layer has
 * one output, followed by tanh when use_tanh (:93-94) and always by the final tanh (:107-108).  W[l] / bias[l] are
 * HOST pointers to the folded (weight-norm applied) row-major (out_dim[l], in_dim[l]) fp32 matrices; ln_weight[l] /
 * ln_bias[l] (out_dim[l] floats) are read for layers with layer_norm[l] only (the arrays may be NULL without any).
 * Limits: n_lin <= HM_MAX_LIN, every width <= 512, latent_dim as for hm_decoder_create.  Such a handle computes in
 * exact fp32 (default) or f16x3 on the matrix cores (hm_decoder_set_precision accepts 0 and 1; f16x3 with the range
 * policy stated there) and is accepted by every entry point that takes an hm_decoder_t; hm_optimize_batch runs it
 * with the render chain as separate launches.  The shipped 8 x 512 / latent_in = [4] models are faster through
 * hm_decoder_create. */
#define HM_MAX_LIN 16
typedef struct hm_decoder_arch {
  int latent_dim;
  int n_lin;
  int use_tanh;
  int in_dim[HM_MAX_LIN];
  int out_dim[HM_MAX_LIN];
  int cat[HM_MAX_LIN];
  int layer_norm[HM_MAX_LIN];
} hm_decoder_arch;
int hm_decoder_create_arch(const hm_decoder_arch* arch, const float* const* W, const float* const* bias,
                           const float* const* ln_weight, const float* const* ln_bias, hm_decoder_t* out);

/* Arithmetic of the decoder GEMMs (the reference computes in fp32, optimizer.py:19):
 *   0  exact fp32 on the f32-input matrix cores (v_mfma_f32_32x32x2_f32; bitwise an fmaf chain)   [default]
 *   1  "f16x3": fp16 matrix cores with hi/lo split operands, three MFMA passes into one fp32 accumulator,
 *      ~2^-22 relative accuracy (fp32 class), 16/3 x the MFMA rate.  Hidden activations and back-propagated
 *      gradients are held as fp16 hi/lo pairs, so they must stay below 65504 in magnitude; a 64-query tile that
 *      exceeds it returns NaN sdf / Jacobian rows (never silent garbage) and hm_optimize_batch ends that instance
 *      with HM_STATUS_SOLVE_FAILED.  Mode 0 has no such limit.  Granularity of the poison = one 64-query tile: the
 *      tiles of the surface points and of the forward ray samples never span two instances, but the render term's
 *      Jacobian samples of ALL instances are packed into one list (one backward-only launch), so a tile there can
 *      hold the last samples of instance b and the first of b + 1: a BACKWARD overflow of one of them then fails
 *      both (the neighbour ends with HM_STATUS_SOLVE_FAILED too; its result is not wrong, it is withheld).  This is
 *      the one exception to "a batched result equals the single-instance result bit for bit"; the drop-in
 *      Optimizer class reruns every failed instance in mode 0, where the statement holds without exception.
 *   2  "f16x3f_f16b" (mixed, NOT fp32-class; BASELINE.json configs[4] "fp16 MFMA decoder"): the forward stages --
 *      residuals, ReLU masks, sdf -- as mode 1, the eight input-gradient (backward) stages in ONE fp16 MFMA pass on the
 *      hi parts: 4 instead of 6 matrix passes per query; Jacobian rows carry ~2^-11 relative rounding per layer.
 *   3  "f16" (plain fp16 MFMA decoder, NOT fp32-class): one fp16 pass for every product, fp16 activations, fp32
 *      accumulation; 128-query tiles (the single activation plane of 128 queries fills the LDS), so each weight byte
 *      serves twice as many queries and only the hi plane is streamed.  sdf, residuals AND Jacobians are fp16-class
 *      (~1e-3 relative); ReLU / occupancy / with-grad decisions can differ from the reference's. */
int hm_decoder_set_precision(hm_decoder_t dec, int precision);
int hm_decoder_get_precision(hm_decoder_t dec);

/* ---- functional parity hooks.
 * mode 0 replaces decode_sdf (wild_completion/utils.py:144-172): d_y[b][i] = sdf(latent_b, pts_b_i).
 * mode 1 replaces get_batch_sdf_jacobian (utils.py:175-193) and the chain rule of compute_sdf_loss
 * (wild_completion/loss.py:219-243): besides d_y, row i of d_J (ldJ >= L+8 floats, ldJ % 4 == 0) receives
 *   [ d sdf/d z (L) | pose part (pose_dim) | .. | sdf ] with the pose part = d sdf/d xyz (3) for pose_dim 0,
 *   d sdf/d xyz . [I | -[p]x] for pose_dim 6 (SE3), ... | p] for pose_dim 7 (Sim3); column L+7 holds sdf.
 * d_pts4: [B][n_stride][4] object-frame points (xyz, w ignored), n_stride % 64 == 0; d_nq[b] = valid points;
 * d_cbias: scratch of 2*B*512 floats. */
int hm_decode_batch(hm_decoder_t dec, int B, const float* d_latent, int ld_latent, const float* d_pts4,
                    const int* d_nq, int n_stride, float* d_cbias, float* d_y, float* d_J, int ldJ,
                    int pose_dim, int mode, void* stream);

/* ---- optimiser.  All keys of the reference's `opt:` YAML block (configs/wild_pepper.yaml:19-57) plus the
 * hidden constants of compute_render_loss (wild_completion/loss.py:11) and the optimiser's depth window
 * (wild_completion/optimizer.py:110). */
typedef struct hm_opt_cfg {
  int scale_on;            /* opt.scale_on: Sim(3) (7 dof) or SE(3) (6 dof)                    */
  int robust_iter;         /* opt.robust_iter                                                   */
  int lm_on, lm_eye;       /* opt.lm.lm_on / lm_eye                                             */
  float lm_lambda_0;       /* opt.lm.lm_lambda_0                                                */
  float s_damp;            /* opt.lm.s_damp                                                     */
  float recon_robust_th;   /* opt.recon.robust_th_m                                             */
  float render_robust_th;  /* opt.render.robust_th_m                                            */
  int n_sample_on_ray;     /* opt.render.n_sample_on_ray (<= 64)                                */
  int log_sdf_occ;         /* opt.render.log_sdf_occ                                            */
  float occ_cutoff;        /* opt.render.occ_cutoff_m                                           */
  int occlusion_on;        /* opt.render.occlusion_on                                           */
  float w_recon, w_depth, w_mask, w_codereg;          /* opt.weight.*                           */
  int max_iter;            /* opt.converge.max_iter                                             */
  float epsilon_g, epsilon_c, epsilon_t, epsilon_r, epsilon_s;   /* opt.converge.*              */
  float occlusion_th;      /* loss.py:11 occlusion_th = 0.03                                    */
  int min_valid_sample;    /* loss.py:11 min_valid_sample = 100                                 */
  float min_grad_thre;     /* loss.py:11 min_grad_thre = 1e-6                                   */
} hm_opt_cfg;

typedef struct hm_limits {
  int max_batch;           /* instances per call                                                */
  int max_points;          /* surface points per instance                                       */
  int max_frames;          /* rendered frames per instance (opt.render.n_frame)                 */
  int max_rays;            /* fg + bg rays per frame                                            */
  int max_samples;         /* samples per ray                                                   */
  int max_grad_samples;    /* Jacobian samples per instance in the render term; 0 = frames*rays*samples */
} hm_limits;

/* One batch of independent fruit instances, everything device resident.  Field <-> reference argument:
 *   points_w     [B][points_stride][3]  `points_w_torch` (optimizer.py:28), n_points[b] valid rows
 *   T_wc         [B][max_frames][16]    render_data["T_wc"][sample_frame_ind] (optimizer.py:77-79,103)
 *   rays         [B][max_frames][max_rays][3]  cat(rays_fg, rays_bg) per frame (optimizer.py:113)
 *   depth        [B][max_frames][max_rays]     cat(depth_fg, depth_bg) per frame (loss.py:26)
 *   n_fg, n_bg   [B][max_frames]        ray counts per frame;  n_frames [B]
 *   latent       [B][L]   in: initial code, out: optimised code (the reference mutates `latent` in place, :248)
 *   T_ow         [B][16]  in: initial object<-world Sim(3), out: optimised
 *   cube_radius  [B]      `cube_radius` (optimizer.py:28);  pose_known [B] (0/1)
 *   iter_count, status [B] outputs.  Instance order is preserved (result b belongs to input b). */
typedef struct hm_batch {
  int B;
  int points_stride;       /* rows between instances in d_points_w; must be <= limits.max_points.  The frame and ray
                              strides of T_wc / rays / depth / n_fg / n_bg are the workspace's max_frames and
                              max_rays (there are no separate fields): pack to the workspace capacities */
  const float* d_points_w;
  const int* d_n_points;
  const float* d_T_wc;
  const float* d_rays;
  const float* d_depth;
  const int* d_n_fg;
  const int* d_n_bg;
  const int* d_n_frames;
  const float* d_cube_radius;
  const int* d_pose_known;
  float* d_latent;
  float* d_T_ow;
  int* d_iter_count;
  int* d_status;
} hm_batch;

/* optional per-iteration debug capture (single-iteration parity tests): damped normal matrix (lower triangle,
 * unknown order [z | pose], leading dimension L+8), right-hand side b and step delta of the LAST executed iteration */
typedef struct hm_debug {
  float* d_A;      /* [B][L+8][L+8] or NULL */
  float* d_b;      /* [B][L+8] or NULL      */
  float* d_delta;  /* [B][L+8] or NULL      */
  int* d_counts;   /* [B][4]: ball-valid samples, Jacobian samples, emitted rays, 0  (last iteration) or NULL */
} hm_debug;

int hm_workspace_create(hm_decoder_t dec, const hm_limits* limits, hm_workspace_t* out);
int hm_workspace_destroy(hm_workspace_t ws);
size_t hm_workspace_bytes(hm_workspace_t ws);

/* Measurement aid: when enabled, hm_optimize_batch brackets every SDF-term decoder launch (the dominant kernel,
 * k_decoder<1,0>) with HIP events on the caller's stream.  hm_workspace_profile_read returns the summed duration
 * [ms] and the number of launches since the last enable; call it only after synchronising the stream. */
int hm_workspace_profile(hm_workspace_t ws, int enable);
int hm_workspace_profile_read(hm_workspace_t ws, double* ms_total, long long* launches);

/* Measurement aid (the whole-iteration algorithmic flop count needs the data-dependent sizes): when
 * enabled (enabling zeroes the sums), every iteration of hm_optimize_batch adds, over the instances active in it,
 *   out5[0] instance-iterations, [1] SDF-term Jacobian queries (N_s; optimizer.py:163-190),
 *   [2] forward-only ray samples K_v (ball-valid samples of the frames that count; loss.py:38-49, optimizer.py:130),
 *   [3] ray samples that need the Jacobian K_g' (loss.py:160-185), [4] emitted rays V (rows of the depth / mask terms)
 * with one extra one-block launch per iteration.  hm_workspace_counters_read synchronises `stream` and copies them. */
int hm_workspace_counters(hm_workspace_t ws, int enable);
int hm_workspace_counters_read(hm_workspace_t ws, long long* out5, void* stream);

/* Replaces Optimizer.shape_pose_joint_opt (mode 0; optimizer.py:28-302) and Optimizer.shape_opt_deepsdf (mode 1;
 * optimizer.py:306-429) for a whole batch.  Instances that converge / become invalid are frozen bit-exactly by
 * device-side flags; results never depend on anything below.
 *
 * Streams.  The work is ordered after everything already on `stream` and before everything the caller enqueues on it
 * afterwards.  A batch of >= 16 instances is cut into instance groups (hm_workspace_set_groups) that run on INTERNAL
 * non-blocking streams (a process-wide pool of at most four per device, shared by all workspaces so that the number of
 * streams -- and with it their mapping onto the runtime's hardware queues -- does not grow with the number of workspaces),
 * forked from and joined back into `stream` by events; the caller's stream sees one fork and one join.  A workspace must
 * not be used by two calls at the same time; calls on DIFFERENT workspaces -- from different host threads, each with its
 * own caller stream; the decoder handle may be shared -- may overlap: a call leases its group streams from the pool for its
 * duration, so concurrent calls get disjoint streams (a call that finds fewer idle streams than it wants groups runs with
 * the groups it can get, down to one group on the caller's stream; results never depend on the grouping).  Two host
 * threads must not enqueue on the SAME caller stream at the same time (the library keeps one launch-scratch block per
 * stream).  `stream` may be a non-blocking stream.
 *
 * Host behaviour -- this call is NOT always a pure enqueue:
 *   - every epsilon_* == 0 (forced iterations): all cfg->max_iter iterations are enqueued and the call returns without
 *     waiting for the device (the early-stop poll is a hipEventQuery, never a wait);
 *   - any epsilon_* > 0 (every shipped YAML): early exits are possible, and with host pacing on (the default) the host
 *     stays at most 3 iterations ahead of the device: from iteration 3 on, each iteration first WAITS
 *     (hipEventSynchronize) for the active-instance count of the iteration three back -- hence also for everything the
 *     caller had queued on `stream` before the call -- and stops enqueueing once that count is zero.  The call then
 *     returns after all but the last <= 3 iterations have EXECUTED.  Why: enqueueing costs ~60 us per iteration, running
 *     one 2 ms and more; an unpaced host has sent all max_iter iterations before the second has finished and the poll
 *     can stop nothing (wild_pepper.yaml: 35 of 50 iterations were empty launches).  Consequences: no host/device overlap
 *     across consecutive calls beyond those last iterations, and the call is ILLEGAL under stream capture
 *     (hipStreamBeginCapture) -- capture with hm_workspace_set_host_pacing(ws, 0).
 * hm_workspace_set_host_pacing(ws, 0) restores the pure enqueue for every cfg: the poll becomes a hipEventQuery, all
 * iterations a finished batch no longer needs are still launched (and find nothing to do: ~50 us each). */
int hm_optimize_batch(hm_workspace_t ws, const hm_opt_cfg* cfg, const hm_batch* batch, int mode,
                      const hm_debug* dbg, void* stream);
/* 1 (default): see above; 0: hm_optimize_batch never waits for the device. */
int hm_workspace_set_host_pacing(hm_workspace_t ws, int on);
/* Instance groups of hm_optimize_batch (each group runs the kernel sequence on its own internal stream, so that one
 * group's under-filled iteration tail shares the chip with another group's main launch; results do not depend on it):
 * 0 = automatic (2 groups from 16 instances on; the default), 1 = everything on the caller's stream, up to 4. */
int hm_workspace_set_groups(hm_workspace_t ws, int groups);

/* Screening of the ray samples under LINEAR occupancy (opt.render.log_sdf_occ = false: lab_pepper.yaml, the challenge
 * config) in the f16x3 arithmetics.  sdf_to_occupancy (wild_completion/utils.py:125-133) clamps, so a sample beyond
 * +-occ_cutoff has occupancy exactly 0 / 1 and no gradient (loss.py:66) whatever its exact sdf.  With screening on, every
 * ball-valid sample first goes through ONE fp16 pass of the decoder; only samples whose fp16 sdf is within
 * occ_cutoff + eps of zero (or not finite) are decoded by the fp32-class f16x3 forward; samples behind the first
 * certainly-inside sample of their ray (transmittance exactly 0) are skipped as well.  Results are bit-identical to the
 * unscreened path as long as eps bounds |sdf_fp16 - sdf_f16x3| (default 1e-3 m: DESIGN.md section 4 has the measurement).
 *   mode 0 off, 1 on (default; ignored for logistic occupancy and for precisions 0 / 3), 2 on + VERIFY: the exact forward
 *   also runs over every sample and each screened-far decision is checked against it; 3 = verify in the FIRST iteration
 *   of each call only (one verified pass per call at 1/max_iter of mode 2's cost); eps <= 0 = default.
 * THE GUARANTEE IS CONDITIONAL: it holds for decoders whose one-pass fp16 error stays inside eps (measured on four
 * decoders; not a property of every network a user may train), and the pass is skipped unless min_grad_thre >= 0 (its
 * dead-sample rule needs `0 > min_grad_thre` to fail as in loss.py:66).  A caller that brings its own decoder should
 * run its first call in mode 2 (or 3) and read `violations`: 0 -> mode 1 is exact for this decoder; otherwise switch to mode 0
 * (or a larger eps) and repeat the call.  The Python optimize_batch / drop-in Optimizer do this once per decoder handle
 * (mode 3: the first iteration of the handle's first screened call).
 * hm_workspace_screening_stats (after enabling hm_workspace_counters, or in mode 2): out4 = {ball-valid samples of valid
 * frames screened, promoted to the exact forward, violations found by mode 2 (must be 0), samples skipped behind a
 * certainly-inside one}; synchronises `stream`; reset != 0 zeroes the sums afterwards (out4 may be NULL). */
int hm_workspace_set_screening(hm_workspace_t ws, int mode, float eps);
int hm_workspace_screening_stats(hm_workspace_t ws, int reset, long long* out4, void* stream);

/* Replaces one call of compute_render_loss per frame (wild_completion/loss.py:8-217) for a batch: runs the render
 * front end + Jacobian pass for the current (latent, T_ow) and leaves, per instance, V[b] depth rows followed (at row
 * offset max_frames*max_rays) by V[b] mask rows of L+8 floats [d res/d z | d res/d pose | res] in d_rows
 * ([B][2*max_frames*max_rays][L+8]); d_V[b] = emitted rays, d_ray_row[b][ray] = row of that ray or -1 (rays are
 * emitted in ascending ray index, fg first: the order torch.unique gives the reference, loss.py:160-166).
 * d_frame_override (optional, [B][max_frames][16]): per frame  T_oc rows 0..2 (12 floats) | d_min | d_max | ball
 * radius | 0, i.e. exactly the arguments `t_obj_cam`, `sampled_ray_depth = linspace(d_min, d_max, M)` and
 * `object_bbx_radius` of compute_render_loss; when NULL they are derived from T_ow and T_wc as the optimiser does
 * (wild_completion/optimizer.py:103-111). */
int hm_render_residuals(hm_workspace_t ws, const hm_opt_cfg* cfg, const hm_batch* batch,
                        const float* d_frame_override, float* d_rows, int* d_V, int* d_ray_row, int* d_counts,
                        void* stream);

/* ---- next row: iso-surface of a decoded SDF grid; replaces convert_sdf_voxels_to_mesh
 * (wild_completion/utils.py:565-588, scikit-image marching cubes on the host) by marching tetrahedra on the GPU over the
 * same grid.  d_sdf [B][n^3] with index (ix*n + iy)*n + iz (the layout create_voxel_grid produces, utils.py:542-562);
 * d_offsets: scratch of B*(n-1)^3 ints; d_tri_count [B]: triangles found (may exceed max_tris: then only cells that fit
 * were emitted); d_tris [B][max_tris][9]: xyz of the three vertices, object frame, grid [-1,1]^3 scaled by cube_radius. */
int hm_extract_surface(int B, const float* d_sdf, int n, float level, float cube_radius, int* d_offsets,
                       int* d_tri_count, float* d_tris, int max_tris, void* stream);
/* the same with marching cubes: vertices exactly the grid-edge crossings (the vertex set of the reference's
 * scikit-image call, utils.py:573); the TRIANGULATION of a cell comes from a table generated at load time (one
 * rule for ambiguous faces, no interior vertices) and may differ from scikit-image's Lewiner tables in ambiguous
 * cells.  At most 8 triangles per cell (size d_tris for 8 * cells in the worst case; a count beyond max_tris is
 * reported in d_tri_count, see above) */
int hm_extract_surface_mc(int B, const float* d_sdf, int n, float level, float cube_radius, int* d_offsets,
                       int* d_tri_count, float* d_tris, int max_tris, void* stream);

/* ---- evaluation primitive: nearest-neighbour distances, the query under ChamferDistance
 * (metrics_3d/chamfer_distance.py:16-26) and PrecisionRecall (metrics_3d/precision_recall.py:13-50), which use an
 * Open3D KD-tree per point.  Exact brute-force scan: d_dist[i] = min_j |a_i - b_j| (Euclidean, not squared).
 * d_a4 [na][4], d_b4 [nb][4]: xyz + one ignored pad float per point; centre both clouds first (fp32 differences). */
int hm_nn_distance(const float* d_a4, int na, const float* d_b4, int nb, float* d_dist, void* stream);

/* ---- caller-side data preparation on the device: the image scans of `get_render_data`
 * (wild_completion/utils.py:39-109).  d_id_imgs [F][H][W] int32 instance-id images, d_depth
 * [F][H][W] float32 depth images of one sequence, resident on the device.
 *
 * hm_prep_stats: for every instance slot s = d_lut[id] (d_lut [lut_size], -1 = not wanted) and frame f,
 *   d_stats[s][f][5] = {count, v_min, v_max, u_min, u_max} of the pixels with that id and depth > 0 (:51-63).  The
 *   caller initialises d_stats to {0, INT_MAX, -1, INT_MAX, -1}.
 * hm_prep_scan: one pass over the padded bounding box of every (instance, frame) pair, raster order (:68-73).
 *   d_pairs [P][8] = {instance id, frame, min_v, max_v, min_u, max_u, n_sel_bg, n_sel_fg}.
 *   gather == 0: d_counts[p] = {#background candidates (id != instance, :74), #foreground candidates (id == instance
 *   and depth > 0, :85)}.  gather == 1: the candidate of raster rank r of kind c (0 bg, 1 fg) is kept when r is in
 *   d_sel[p][c][0..n_sel) (sorted ascending; the host's np.random.choice draw, :79/:90) and written to output row
 *   d_perm[p][c][i] (the position of r in the draw), or, when n_sel < 0 (no sub-sampling: at most `cap` candidates),
 *   to row r: d_pix [P][2][cap][2] = (u, v), d_depth_out [P][2][cap], d_rays [P][2][cap][3] = K^-1 [u, v, 1] rounded
 *   from fp64 like get_rays (:23-37); d_invK [9] fp64 row-major. */
int hm_prep_stats(const int* d_id_imgs, const float* d_depth, int F, int H, int W, const int* d_lut, int lut_size,
                  int B, int* d_stats, void* stream);
int hm_prep_scan(const int* d_id_imgs, const float* d_depth, int H, int W, const int* d_pairs, int P, int gather,
                 int* d_counts, const int* d_sel, const int* d_perm, int cap, const double* d_invK, int* d_pix,
                 float* d_depth_out, float* d_rays, void* stream);


/* hm_prep_dbscan: `clean_pcd` (wild_completion/utils.py:407-417) -- DBSCAN of each instance's surface samples, one
 *   workgroup per instance.  d_pts [B][n_stride][3] fp64, d_n_pts [B], n_stride <= 5120, eps = cluster_dist_thre,
 *   d_min_pts [B] = max(1, int(n * outlier_point_ratio)).  d_comp [B][n_stride]: the smallest core-point index of the
 *   point's cluster, -1 for noise; a border point joins the adjacent cluster with the smallest first core index (what an
 *   index-ordered DBSCAN expansion does).  Ranking the distinct values gives the DBSCAN labels.
 * hm_prep_box_select: the background crop of `get_pose_init` (:442-447): indices, in input order, of the points of
 *   d_pts [n][3] fp64 inside each closed box d_boxes [B][6] = {min xyz, max xyz}.  gather == 0: d_counts [B];
 *   gather == 1: indices of box b written from d_idx + d_offsets[b]. */
int hm_prep_dbscan(const double* d_pts, const int* d_n_pts, int n_stride, int B, double eps, const int* d_min_pts,
                   int* d_comp, void* stream);
int hm_prep_box_select(const double* d_pts, int n, const double* d_boxes, int B, int gather, int* d_counts,
                       const long long* d_offsets, int* d_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif
