// Stand-alone consumer of libhortihip.so: no Python, no torch -- only the public header, the HIP runtime for the
// caller-owned device buffers, and the C ABI.  Builds an analytic decoder (every hidden unit i < 3 of layer 0 copies one
// coordinate, the rest of the network passes it on), decodes a handful of points in both arithmetics and checks the
// closed form  sdf = tanh(w * relu(x + y + z... ))  -- see the comments below -- and runs one Levenberg-Marquardt iteration of
// hm_optimize_batch (shape-only mode) against ITS closed form.  Prints ABI_SMOKE_OK on success.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <string.h>
#include <vector>
#include "hortimapping_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define HM(x) do { int r_ = (x); if (r_ != 0) { printf("hm error %d at %d: %s\n", r_, __LINE__, hm_last_error()); return 3; } } while (0)

int main() {
  const int L = 32, H = 512, D0 = L + 3, m = H - D0;
  // lin0: h_i = relu(c_i) for i < 3 (c = x, y, z), h_3 = relu(z_0) (first latent), others 0
  std::vector<float> W0((size_t)H * D0, 0.f), W1((size_t)H * H, 0.f), W3((size_t)m * H, 0.f), W4((size_t)H * H, 0.f),
      W8(H, 0.f), bz(H, 0.f), b8(1, 0.1f);
  for (int c = 0; c < 3; ++c) W0[(size_t)c * D0 + L + c] = 1.f;
  W0[(size_t)3 * D0 + 0] = 1.f;
  for (int i = 0; i < H; ++i) W1[(size_t)i * H + i] = 1.f;          // identity: lin1, lin2, lin5..7
  for (int i = 0; i < m; ++i) W3[(size_t)i * H + i] = 1.f;          // lin3 keeps units 0..m-1
  for (int i = 0; i < m; ++i) W4[(size_t)i * H + i] = 1.f;          // lin4 passes h3 through, ignores the skip input
  W8[0] = 0.5f; W8[1] = 0.25f; W8[2] = -1.f; W8[3] = 2.f;           // sdf = tanh(.5 x+ + .25 y+ - z+ + 2 z0+ + .1)
  const float* Ws[9] = {W0.data(), W1.data(), W1.data(), W3.data(), W4.data(), W1.data(), W1.data(), W1.data(), W8.data()};
  std::vector<float> b3(m, 0.f);
  const float* bs[9] = {bz.data(), bz.data(), bz.data(), b3.data(), bz.data(), bz.data(), bz.data(), bz.data(), b8.data()};
  hm_decoder_t dec = nullptr;
  HM(hm_decoder_create(L, Ws, bs, &dec));
  if (hm_decoder_latent_dim(dec) != L) { printf("latent dim\n"); return 4; }

  const int B = 2, NS = 64, ldJ = L + 8;
  std::vector<float> lat((size_t)B * L, 0.f), pts((size_t)B * NS * 4, 0.f);
  lat[0] = 0.3f; lat[L] = -0.7f;                                    // z0 of instance 0 / 1
  int nq_h[B] = {5, 3};
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < nq_h[b]; ++i) {
      float* p = &pts[((size_t)b * NS + i) * 4];
      p[0] = 0.1f * (i + 1) * (b ? -1.f : 1.f); p[1] = 0.05f * i; p[2] = 0.02f * (i - 1);
    }
  float *d_lat, *d_pts, *d_cb, *d_y, *d_J; int* d_nq;
  CK(hipMalloc(&d_lat, lat.size() * 4)); CK(hipMalloc(&d_pts, pts.size() * 4)); CK(hipMalloc(&d_cb, 2 * B * 512 * 4));
  CK(hipMalloc(&d_y, B * NS * 4)); CK(hipMalloc(&d_J, (size_t)B * NS * ldJ * 4)); CK(hipMalloc(&d_nq, B * 4));
  CK(hipMemcpy(d_lat, lat.data(), lat.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_nq, nq_h, B * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int prec = 0; prec < 2; ++prec) {
    HM(hm_decoder_set_precision(dec, prec));
    CK(hipMemsetAsync(d_J, 0, (size_t)B * NS * ldJ * 4, st));
    HM(hm_decode_batch(dec, B, d_lat, L, d_pts, d_nq, NS, d_cb, d_y, d_J, ldJ, 7, 1, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> y(B * NS), J((size_t)B * NS * ldJ);
    CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(J.data(), d_J, J.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < nq_h[b]; ++i) {
        const float* p = &pts[((size_t)b * NS + i) * 4];
        auto rl = [](float v) { return v > 0.f ? v : 0.f; };
        const float z0 = lat[(size_t)b * L];
        const float a = 0.5f * rl(p[0]) + 0.25f * rl(p[1]) - rl(p[2]) + 2.f * rl(z0) + 0.1f;
        const float want = std::tanh(a), dy = 1.f - want * want;
        const float gx = dy * 0.5f * (p[0] > 0), gz0 = dy * 2.f * (z0 > 0);
        const float* row = &J[((size_t)b * NS + i) * ldJ];
        if (std::fabs(y[b * NS + i] - want) > 2e-6f || std::fabs(row[L + 7] - want) > 2e-6f ||
            std::fabs(row[L] - gx) > 2e-6f || std::fabs(row[0] - gz0) > 2e-6f) {
          printf("mismatch prec %d b %d i %d: y %.8f want %.8f dx %.8f want %.8f dz0 %.8f want %.8f\n", prec, b, i,
                 y[b * NS + i], want, row[L], gx, row[0], gz0);
          return 5;
        }
      }
  }
  // a refused request must come back as an error code with a message, not a crash
  if (hm_decode_batch(dec, B, d_lat, L, d_pts, d_nq, 63, d_cb, d_y, d_J, ldJ, 7, 1, st) == 0) { printf("stride 63 accepted\n"); return 6; }
  if (hm_last_error() == nullptr || hm_last_error()[0] == 0) { printf("no error text\n"); return 7; }
  // ---- hm_decoder_create_arch: the same closed-form function as a 64 - 64 - 1 table (three Linear layers, no skip) ----
  {
    const int Hs = 64;
    std::vector<float> A0((size_t)Hs * D0, 0.f), A1((size_t)Hs * Hs, 0.f), A2(Hs, 0.f), z64(Hs, 0.f);
    for (int c = 0; c < 3; ++c) A0[(size_t)c * D0 + L + c] = 1.f;
    A0[(size_t)3 * D0 + 0] = 1.f;
    for (int i = 0; i < Hs; ++i) A1[(size_t)i * Hs + i] = 1.f;
    A2[0] = 0.5f; A2[1] = 0.25f; A2[2] = -1.f; A2[3] = 2.f;
    hm_decoder_arch arch;
    memset(&arch, 0, sizeof(arch));
    arch.latent_dim = L; arch.n_lin = 3;
    arch.in_dim[0] = D0; arch.out_dim[0] = Hs; arch.in_dim[1] = Hs; arch.out_dim[1] = Hs; arch.in_dim[2] = Hs; arch.out_dim[2] = 1;
    const float* Wa[HM_MAX_LIN] = {A0.data(), A1.data(), A2.data()};
    const float* ba[HM_MAX_LIN] = {z64.data(), z64.data(), b8.data()};
    hm_decoder_t dany = nullptr;
    HM(hm_decoder_create_arch(&arch, Wa, ba, nullptr, nullptr, &dany));
    if (hm_decoder_set_precision(dany, 3) == 0) { printf("plain fp16 accepted by an any-architecture decoder\n"); return 6; }
    for (int prec = 0; prec < 2; ++prec) {
    HM(hm_decoder_set_precision(dany, prec));
    CK(hipMemsetAsync(d_J, 0, (size_t)B * NS * ldJ * 4, st));
    HM(hm_decode_batch(dany, B, d_lat, L, d_pts, d_nq, NS, d_cb, d_y, d_J, ldJ, 7, 1, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> y(B * NS), J((size_t)B * NS * ldJ);
    CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(J.data(), d_J, J.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < nq_h[b]; ++i) {
        const float* p = &pts[((size_t)b * NS + i) * 4];
        auto rl = [](float v) { return v > 0.f ? v : 0.f; };
        const float z0 = lat[(size_t)b * L];
        const float want = std::tanh(0.5f * rl(p[0]) + 0.25f * rl(p[1]) - rl(p[2]) + 2.f * rl(z0) + 0.1f), dy = 1.f - want * want;
        const float* row = &J[((size_t)b * NS + i) * ldJ];
        if (std::fabs(y[b * NS + i] - want) > 2e-6f || std::fabs(row[L + 7] - want) > 2e-6f ||
            std::fabs(row[L] - dy * 0.5f * (p[0] > 0)) > 2e-6f || std::fabs(row[0] - dy * 2.f * (z0 > 0)) > 2e-6f) {
          printf("any-architecture mismatch prec %d b %d i %d: y %.8f want %.8f\n", prec, b, i, y[b * NS + i], want);
          return 5;
        }
      }
    }
    arch.out_dim[1] = 63;                        // a table that does not chain must be refused with a message
    hm_decoder_t bad = nullptr;
    if (hm_decoder_create_arch(&arch, Wa, ba, nullptr, nullptr, &bad) == 0) { printf("bad table accepted\n"); return 6; }
    HM(hm_decoder_destroy(dany));
  }
  // ---- hm_optimize_batch (mode 1 = Optimizer.shape_opt_deepsdf, optimizer.py:306-429) with a closed-form answer ----
  // Points on the +x axis (y = z = 0), T_ow = identity, latent z0 > 0: sdf_i = tanh(0.5 x_i + 2 z0 + 0.1), only z0 has a
  // non-zero Jacobian column J_i = 2 (1 - sdf_i^2).  One LM iteration (optimizer.py:362-397):
  //   H = mean(J^2) + w_c,  b = -mean(J sdf) - w_c z0,  H += lambda_0 H,  z0 += b / H;   every other code entry stays 0.
  {
    HM(hm_decoder_set_precision(dec, 0));
    const int NP = 100;
    hm_limits lim = {1, NP, 0, 0, 0, 0};
    hm_workspace_t ws = nullptr;
    HM(hm_workspace_create(dec, &lim, &ws));
    hm_opt_cfg cfg = {};
    cfg.scale_on = 1; cfg.robust_iter = 5; cfg.lm_on = 1; cfg.lm_eye = 0; cfg.lm_lambda_0 = 0.1f; cfg.s_damp = 1e-3f;
    cfg.recon_robust_th = 0.01f; cfg.render_robust_th = 0.05f; cfg.n_sample_on_ray = 16; cfg.log_sdf_occ = 1;
    cfg.occ_cutoff = 0.01f; cfg.occlusion_on = 1; cfg.w_recon = 1.f; cfg.w_depth = 0.05f; cfg.w_mask = 5e-4f;
    cfg.w_codereg = 5e-4f; cfg.max_iter = 1; cfg.occlusion_th = 0.03f; cfg.min_valid_sample = 100; cfg.min_grad_thre = 1e-6f;
    std::vector<float> pw((size_t)NP * 3, 0.f), z(L, 0.f), T(16, 0.f);
    for (int i = 0; i < NP; ++i) pw[(size_t)i * 3] = 0.01f * (i + 1);
    z[0] = 0.3f;
    T[0] = T[5] = T[10] = T[15] = 1.f;
    float *d_pw, *d_z, *d_T; int *d_np, *d_it, *d_st;
    CK(hipMalloc(&d_pw, pw.size() * 4)); CK(hipMalloc(&d_z, L * 4)); CK(hipMalloc(&d_T, 64));
    CK(hipMalloc(&d_np, 4)); CK(hipMalloc(&d_it, 4)); CK(hipMalloc(&d_st, 4));
    CK(hipMemcpy(d_pw, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_z, z.data(), L * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_T, T.data(), 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_np, &NP, 4, hipMemcpyHostToDevice));
    hm_batch bt = {};
    bt.B = 1; bt.points_stride = NP; bt.d_points_w = d_pw; bt.d_n_points = d_np; bt.d_latent = d_z; bt.d_T_ow = d_T;
    bt.d_iter_count = d_it; bt.d_status = d_st;
    HM(hm_optimize_batch(ws, &cfg, &bt, 1, nullptr, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> zo(L);
    int it = -1, stt = -1;
    CK(hipMemcpy(zo.data(), d_z, L * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&it, d_it, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&stt, d_st, 4, hipMemcpyDeviceToHost));
    double sJJ = 0, sJr = 0;
    for (int i = 0; i < NP; ++i) {
      const double y = std::tanh(0.5 * (double)pw[(size_t)i * 3] + 2.0 * 0.3 + 0.1), J = 2.0 * (1.0 - y * y);
      sJJ += J * J; sJr += J * y;
    }
    const double Hd = (sJJ / NP + 5e-4) * 1.1, bd = -sJr / NP - 5e-4 * 0.3, want = 0.3 + bd / Hd;
    if (it != 1 || stt != HM_STATUS_MAX_ITER || std::fabs(zo[0] - want) > 2e-6 * std::fabs(want)) {
      printf("optimize mismatch: iter %d status %d z0 %.8f want %.8f\n", it, stt, zo[0], want);
      return 8;
    }
    for (int i = 1; i < L; ++i) if (zo[i] != 0.f) { printf("latent entry %d moved: %g\n", i, zo[i]); return 9; }
    // an instance that exceeds the packed stride is refused per instance, not truncated
    const int too_many = NP + 1;
    CK(hipMemcpy(d_np, &too_many, 4, hipMemcpyHostToDevice));
    HM(hm_optimize_batch(ws, &cfg, &bt, 1, nullptr, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(&stt, d_st, 4, hipMemcpyDeviceToHost));
    if (stt != HM_STATUS_LIMIT) { printf("oversized instance not flagged: status %d\n", stt); return 10; }
    HM(hm_workspace_destroy(ws));
  }
  HM(hm_decoder_destroy(dec));
  {
    // data-preparation entry points: box crop of a point cloud (get_pose_init) and DBSCAN of two well-separated blobs
    const int n = 1000;
    std::vector<double> pts((size_t)n * 3);
    for (int i = 0; i < n; ++i) { pts[3 * i] = 0.001 * i; pts[3 * i + 1] = 0.5; pts[3 * i + 2] = (i % 2) ? 0.1 : 0.9; }
    const double box[6] = {0.1995, 0.0, 0.0, 0.3005, 1.0, 0.5};           // x in [0.2, 0.3], z = 0.1 only: odd i of 200..300
    double *d_pts, *d_box; int *d_cnt, *d_idx; long long* d_off;
    if (hipMalloc(&d_pts, pts.size() * 8) != hipSuccess || hipMalloc(&d_box, 48) != hipSuccess || hipMalloc(&d_cnt, 4) != hipSuccess ||
        hipMalloc(&d_idx, n * 4) != hipSuccess || hipMalloc(&d_off, 8) != hipSuccess) return 4;
    const long long zero = 0;
    if (hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_box, box, 48, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_off, &zero, 8, hipMemcpyHostToDevice) != hipSuccess) return 4;
    HM(hm_prep_box_select(d_pts, n, d_box, 1, 0, d_cnt, d_off, d_idx, nullptr));
    int cnt = -1;
    if (hipMemcpy(&cnt, d_cnt, 4, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    HM(hm_prep_box_select(d_pts, n, d_box, 1, 1, d_cnt, d_off, d_idx, nullptr));
    std::vector<int> idx(n, -1);
    if (hipMemcpy(idx.data(), d_idx, (size_t)cnt * 4, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    if (cnt != 50 || idx[0] != 201 || idx[49] != 299) { printf("box select: %d %d %d\n", cnt, idx[0], idx[49]); return 5; }
    std::vector<double> blobs((size_t)64 * 3);
    for (int i = 0; i < 64; ++i) { blobs[3 * i] = (i < 40 ? 0.0 : 1.0) + 0.001 * (i % 8); blobs[3 * i + 1] = 0.001 * (i / 8); blobs[3 * i + 2] = 0.0; }
    double* d_b; int *d_n, *d_mp, *d_comp;
    const int nb = 64, mp = 3;
    if (hipMalloc(&d_b, blobs.size() * 8) != hipSuccess || hipMalloc(&d_n, 4) != hipSuccess || hipMalloc(&d_mp, 4) != hipSuccess ||
        hipMalloc(&d_comp, 64 * 4) != hipSuccess) return 4;
    if (hipMemcpy(d_b, blobs.data(), blobs.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_n, &nb, 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_mp, &mp, 4, hipMemcpyHostToDevice) != hipSuccess) return 4;
    HM(hm_prep_dbscan(d_b, d_n, 64, 1, 0.01, d_mp, d_comp, nullptr));
    std::vector<int> comp(64);
    if (hipMemcpy(comp.data(), d_comp, 64 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    for (int i = 0; i < 64; ++i)
      if (comp[i] != (i < 40 ? 0 : 40)) { printf("dbscan: point %d in cluster %d\n", i, comp[i]); return 5; }
    (void)hipFree(d_pts); (void)hipFree(d_box); (void)hipFree(d_cnt); (void)hipFree(d_idx); (void)hipFree(d_off);
    (void)hipFree(d_b); (void)hipFree(d_n); (void)hipFree(d_mp); (void)hipFree(d_comp);
  }
  printf("ABI_SMOKE_OK\n");
  return 0;
}
