/* libhortihip.so -- test, A/B and performance-analysis hooks.  NOT part of the drop-in surface (include/hortimapping_amd.h):
 * nothing a caller of the optimiser needs, nothing the reference has a counterpart for.  Kept in their own header so that
 * the product ABI carries no process-global switches (round-3 review).
 *
 * Scope of the switches: hm_workspace_set_debug() sets them for ONE workspace; hm_debug_split_render() /
 * hm_debug_force_direct_solve() set the process default that workspaces without an override follow.  Either way the value
 * is read ONCE, when hm_optimize_batch / hm_render_residuals is entered, so a call never sees a switch change under it.
 * The trace registrations (hm_debug_set_*trace*) are process-wide and meant for single-threaded analysis scripts. */
#ifndef HORTIMAPPING_AMD_DEBUG_H
#define HORTIMAPPING_AMD_DEBUG_H

#include "hortimapping_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* split_render: 1 = the f16x3 render chain as separate launches with a forward+backward Jacobian pass (round 2) instead
 * of the fused grid + backward-only pass from saved ReLU masks (same bits either way); force_direct_solve: 1 = every solve
 * takes the blocked-Cholesky fallback of the solve kernel.  -1 = follow the process default. */
int hm_workspace_set_debug(hm_workspace_t ws, int split_render, int force_direct_solve);
/* (hm_workspace_set_groups moved to the product header in round 5: the drop-in Optimizer and bench.py use it.) */
/* Normal equations of the f16x3 arithmetics: 0 (default) = the fp32-input kernel of rounds 1-4 (what exact f32 always runs);
 * 1 = K4h, fp16 matrix cores on split operands (hm_normal_eq.hip): +1.7 % on C2-joint, H and b within 1e-7 ... 7e-7 of the
 * fp32-input kernel -- opt-in, because that difference is enough to move a Jacobian-sample count of one eight-iteration
 * free-pose trajectory off the oracle's (tests/test_gpu_configs.py::test_frame_turns_invalid_mid_trajectory_L256), which the
 * fp32-input kernel reproduces exactly; 2 = in -DHM_EXPERIMENTAL builds the whole-instance kernel K4w (measured slower end
 * to end), elsewhere the same as 1.  A/B and the test that they agree to fp32 rounding. */
int hm_workspace_set_k4_split(hm_workspace_t ws, int on);

/* ---- performance-analysis aids (not part of the drop-in surface): when a device buffer is registered, block 0 of
 * the f16x3 decoder kernel / of the solve kernel writes shader-clock stamps per stage into it (scripts/gpu_trace_*.py). */
void hm_debug_set_trace(long long* d_buf);      /* [NSTAGE * 8 + 1] or NULL (the product kernel writes [NSTAGE * 4 + 1] of it) */
void hm_debug_set_k5_trace(long long* d_buf);   /* [32] or NULL */
void hm_debug_force_direct_solve(int on);        /* tests: 1 = every solve takes the blocked-Cholesky fallback of K5 */
void hm_debug_set_trace_thread(int tid);         /* which thread of workgroup 0 writes the hm_debug_set_trace stamps */
void hm_debug_split_render(int on);              /* A/B + tests: 1 = the f16x3 render chain as separate launches with a
                                                  * forward+backward Jacobian pass (round 2) instead of the fused grid +
                                                  * backward-only pass from saved ReLU masks; same bits either way */
void hm_debug_set_k1p_trace(long long* d_buf);  /* [5 * 16] or NULL: per-stage stamps of the plain-fp16 decoder kernel */

/* ---- unit hooks (tests): the device functions of the solve kernel / normal-equation kernel on caller-supplied values.
 * hm_debug_exp_map replaces exp_sim3 (sim3 != 0; wild_completion/utils.py:279-324) / exp_se3 (:220-254) for n tangents
 * [n][7] (translation, rotation, log-scale; the 7th entry is ignored for se3) -> [n][16] row-major 4x4.
 * hm_debug_huber replaces get_robust_res (utils.py:343-358): d_rho[i] = w_i^2, d_robust_res[i] = w_i r_i (optional). */
int hm_debug_exp_map(const float* d_tangents, int n, int sim3, float* d_T, void* stream);
int hm_debug_huber(const float* d_res, int n, float threshold, float* d_rho, float* d_robust_res, void* stream);

#ifdef __cplusplus
}
#endif
#endif
