// Shared device/host declarations of libhortihip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace hm {

constexpr int HID = 512;        // hidden width of the DeepSDF decoder (specs.json:8)
constexpr int TQ = 64;          // decoder queries per workgroup tile (2 MFMA N-blocks of 32)
constexpr int NWAVE = 8;        // waves per decoder workgroup
constexpr int NSTAGE = 16;      // 8 forward GEMM stages (lin0..lin7; lin8 is a VALU dot) + 8 backward
constexpr int POSE_PAD = 8;     // pose columns reserved in a Jacobian row
// Default margin [m] of the linear-occupancy screening (hm_render.hip k_promote): bound on |sdf_fp16 - sdf_f16x3| with
// room to spare.  Measured (scripts/measure_screen_eps.py, profiles/r05_screen_eps.txt): see DESIGN.md section 4.
constexpr float HM_SCREEN_EPS_DEFAULT = 1.0e-3f;
constexpr int MAX_L = 256;

// epilogue kinds of a decoder stage
enum {
  EPI_FWD = 0,    // bias + ReLU, record mask, write activations
  EPI_FWD3 = 1,   // lin3: as FWD, rows m..m+2 are overwritten with xyz (input of the skip layer)
  EPI_FWD7 = 2,   // lin7: as FWD; followed by the lin8 dot + tanh
  EPI_BWD = 3,    // multiply by ReLU mask of the producing layer, write gradients
  EPI_BWD4 = 4,   // transpose of lin4: h3 part masked+written, latent part kept in registers
  EPI_BWD0 = 5,   // transpose of lin0 (latent columns): accumulate onto the kept latent part, emit
};

struct StageDesc {
  const float* wp;     // packed A operand: [(mb - mb_lo)][kg][64 lanes][4]
  const float* bias;   // per-output bias (fwd) or nullptr; per-instance stages use c0/c4 instead
  int n_kg;            // K groups of 8
  int mb_lo, mb_hi;    // valid 32-row output blocks [lo, hi)
  int epi;
  int layer;           // forward layer whose mask is written (fwd) or read (bwd)
  int inst_bias;       // 0: bias ptr, 1: c0[b], 2: c4[b]
};

// f16x3 variant of a stage: A operands split into fp16 hi/lo at a per-stage power-of-two scale
struct StageDescH {
  const void* wp;      // packed [(mb - mb_lo)][k16 step][hi|lo][64 lanes][8 halves]
  int n_k16;           // K steps of 16
  float unscale;       // 2^-shift: accumulators hold 2^shift * (W X)
  int mb_stride;       // distance between row blocks in 16-byte units (n_k16 * 128 + pad)
};

struct DecoderDev {
  int L, m, m_pad, mb_zx;        // m = 509 - L, m_pad = 512 - L, mb_zx = 16 - L/32
  StageDesc st[NSTAGE];
  StageDescH sth[NSTAGE];
  const float* w8;               // [512]
  float b8;
  const float* w0x;              // [512][4]  xyz columns of lin0 (4th = 0)
  const float* w4x;              // [512][4]  xyz columns of lin4
  const float* w0z;              // [512][L]  latent columns of lin0 (row-major) for the per-instance bias
  const float* w4z;              // [512][L]
  const float* b0;               // [512]
  const float* b4;               // [512]
  // plain-fp16 kernel (hm_decoder_p.hip): per-wave weight streams (hm_pack.hip pack_stream_p)
  const void* ps;                // [8 waves][ps_steps][2 row blocks][64 lanes][8 halves]
  int ps_steps;                  // K-steps of 16 per wave stream (every stage padded to groups of 4, + slack)
  int pgrp[NSTAGE];              // groups of 4 K-steps per stage
  float pus[NSTAGE];             // 2^-shift per stage: accumulators hold 2^shift * (W X)
  int pswap[NSTAGE];             // 1: slot 0 of wave w is block w + 8, slot 1 block w (stages with no valid block below 8)
};

// ---- any-architecture decoder (hm_decoder_any.hip): the layer table of deepsdf/networks/deep_sdf_decoder.py:29-72 ----
#define HM_ANY_MAX_LIN 16       // Linear layers (== HM_MAX_LIN of include/hortimapping_amd.h)
#define HM_ANY_MAX_WIDTH 512    // widest layer input / output

struct AnyLayer {
  const float* wf;     // forward A operand  (W,   rows = out_dim, K = in_dim),  pack_stage order, zero padded
  const float* wb;     // backward A operand (W^T, rows = in_dim,  K = out_dim), pack_stage order, zero padded
  const float* bias;   // [512], zero padded
  const float* gamma;  // [512] LayerNorm weight (layers with ln), zero padded
  const float* beta;   // [512] LayerNorm bias
  // f16x3 copies (pack_stage_h: fp16 hi / lo planes at a per-operand power-of-two scale, K padded to steps of 16)
  const void* wfh;     // forward A operand
  const void* wbh;     // backward A operand
  int kf16, kb16;      // K steps of 16: ceil(in_dim / 16), ceil(out_dim / 16)
  float usf, usb;      // 2^-shift of the two operands: accumulators hold 2^shift * (W X)
  int in_dim, out_dim;
  int cat;             // input of this layer: 0 = previous output, 1 = cat[., z, xyz] (latent_in), 2 = cat[., xyz] (xyz_in_all)
  int ln;              // LayerNorm between this Linear and its ReLU
};

struct AnyDev {
  int L, n_lin, use_tanh, n_ln;
  AnyLayer lay[HM_ANY_MAX_LIN];
};

}  // namespace hm

struct hm_decoder_s {
  hm::DecoderDev dev;
  int precision;       // 0: exact fp32 MFMA, 1: fp16 hi/lo split (3 fp16 MFMA passes, ~2^-22 relative),
                       // 2: as 1 in the forward stages, ONE fp16 pass in the backward stages (Jacobians ~1e-3),
                       // 3: plain fp16 MFMA everywhere (hm_decoder_p.hip, 128-query tiles; fp16-class results)
  void* d_blob;        // one allocation holding every packed array
  size_t blob_bytes;
  int L;
  int generic;         // 1: built by hm_decoder_create_arch: `any` is valid, `dev` is not; precisions 0 (exact fp32) and 1 (f16x3)
  hm::AnyDev any;
};

#define HM_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      hm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                                  \
    }                                                                             \
  } while (0)

extern "C" void hm_set_error(const char* fmt, ...);
