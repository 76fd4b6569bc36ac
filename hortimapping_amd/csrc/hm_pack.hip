// Host side of the decoder handle: weight ingest and packing into MFMA A-operand order.
//
// The reference rebuilds the decoder from specs.json + a state dict (deepsdf/deep_sdf/workspace.py:203-225,
// deepsdf/networks/deep_sdf_decoder.py:29-72).  Here the caller passes the nine *folded* row-major fp32
// matrices (weight-norm already applied: W = g * v / ||v||) and biases; we pre-pack each GEMM stage as
// [row block of 32][K group of 8][lane 0..63][4] so that one global_load_dwordx4 per lane yields the A
// operands of four consecutive v_mfma_f32_32x32x2_f32 (lane l: row l&31, k = 8*kg + 4*(l>>5) + j).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <map>
#include <mutex>
#include <vector>

#include <hip/hip_fp16.h>
#include <math.h>

#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

static thread_local char g_err[512] = "";

extern "C" void hm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hm_last_error(void) { return g_err; }

namespace hm {
int scratch_get(void** p, size_t bytes, hipStream_t stream) {
  struct Block { void* ptr; size_t bytes; };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Block> blocks;
  int dev = 0;
  HM_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  Block& b = blocks[std::make_pair(dev, stream)];
  if (b.bytes < bytes) {
    // grow (rare: a stream's first launch, or a larger launch than any before; geometric, so a few times per stream at
    // most): launches already enqueued on this stream may still use the old block -> wait for them, then release it
    size_t want = bytes > b.bytes + b.bytes / 2 ? bytes : b.bytes + b.bytes / 2;
    want = (want + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
    if (b.ptr != nullptr) {
      HM_CHECK_HIP(hipStreamSynchronize(stream));
      HM_CHECK_HIP(hipFree(b.ptr));
    }
    b.ptr = nullptr; b.bytes = 0;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) { hm_set_error("hipMalloc(%zu) of a launch scratch block failed: %s", want, hipGetErrorString(e)); return -2; }
    b.ptr = q; b.bytes = want;
  }
  *p = b.ptr;
  return 0;
}
}  // namespace hm

namespace {

struct Blob {
  std::vector<float> host;
  size_t alloc(size_t n) {           // returns offset (in floats), 64-byte aligned
    size_t off = (host.size() + 15) & ~size_t(15);
    host.resize(off + n, 0.f);
    return off;
  }
};

// A stage's operand matrix, materialised ONCE from its accessor (512 rows x `cols`, zero outside the accessor's range): the
// packers below read it three to five times per element, and through a std::function that was most of the 0.4 s a decoder
// handle took to create (round 6; the packed bytes are the same).
struct Dense {
  std::vector<float> v;
  int cols = 0;
  float operator()(int r, int c) const { return c < cols ? v[(size_t)r * cols + c] : 0.f; }
};
Dense materialise(const std::function<float(int, int)>& A, int mb_lo, int mb_hi, int kmax, int cols) {
  Dense d;
  d.cols = cols;
  d.v.assign((size_t)HID * cols, 0.f);
  for (int r = mb_lo * 32; r < mb_hi * 32; ++r)
    for (int c = 0; c < kmax; ++c) d.v[(size_t)r * cols + c] = A(r, c);
  return d;
}

// A(r, c) accessor -> packed [n_mb][n_kg][64][4]
size_t pack_stage(Blob& blob, int mb_lo, int mb_hi, int n_kg, const Dense& A) {
  const int n_mb = mb_hi - mb_lo;
  size_t off = blob.alloc((size_t)n_mb * n_kg * 256);
  for (int mbi = 0; mbi < n_mb; ++mbi)
    for (int kg = 0; kg < n_kg; ++kg)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int r = (mb_lo + mbi) * 32 + (lane & 31);
          const int c = kg * 8 + (lane >> 5) * 4 + j;
          blob.host[off + (((size_t)mbi * n_kg + kg) * 64 + lane) * 4 + j] = A(r, c);
        }
  return off;
}

// fp16 hi/lo packing of a stage for the f16x3 kernel.  With shift s (2^s * max|A| < 32768):
//   hi = fp16(A * 2^s),  lo = fp16(A * 2^s - hi);   the kernel forms  hi*Xh + (hi*2^-11)*(Xl*2^11) + lo*Xh.
// Layout [mb][k16][hi|lo][lane][8]: lane l holds row l&31, k = 16*step + 8*(l>>5) + j.
size_t pack_stage_h(std::vector<uint16_t>& blob, int mb_lo, int mb_hi, int n_k16, int mb_stride,
                    const Dense& A, float* unscale) {
  const int n_mb = mb_hi - mb_lo;
  float mx = 0.f;
  for (int r = mb_lo * 32; r < mb_hi * 32; ++r)
    for (int c = 0; c < n_k16 * 16; ++c) mx = fmaxf(mx, fabsf(A(r, c)));
  int shift = 12;
  while (shift > -12 && ldexpf(mx, shift) >= 32768.f) --shift;
  *unscale = ldexpf(1.f, -shift);
  size_t off = (blob.size() + 63) & ~size_t(63);
  blob.resize(off + (size_t)n_mb * mb_stride * 8, 0);
  for (int mbi = 0; mbi < n_mb; ++mbi)
    for (int ks = 0; ks < n_k16; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int r = (mb_lo + mbi) * 32 + (lane & 31);
          const int c = ks * 16 + (lane >> 5) * 8 + j;
          const float v = ldexpf(A(r, c), shift);
          const __half hi = __float2half_rn(v);
          const __half lo = __float2half_rn(v - __half2float(hi));
          const size_t base = off + ((size_t)mbi * mb_stride + ((size_t)ks * 2) * 64 + lane) * 8 + j;
          blob[base] = *reinterpret_cast<const uint16_t*>(&hi);
          blob[base + 64 * 8] = *reinterpret_cast<const uint16_t*>(&lo);
        }
  return off;
}


// Weight streams of the plain-fp16 kernel (hm_decoder_p.hip).  Each of the 8 waves of a workgroup reads ONE contiguous
// stream in exactly the order it consumes it: [stage][K-step of 16][row block r = 0, 1 -> 32-row block w + 8 r][lane][8
// halves], the step count of every stage padded to a multiple of 4 with zero steps (the kernel's ring of four operand
// sets then has the same phase at every stage entry and its prefetch runs across stage boundaries without any address
// logic).  fp16(A * 2^shift) at the per-stage power-of-two shift of pack_stage_h.  K order inside a step: lane half h holds
// k = 16 t + 4 h + {0..3} and 16 t + 8 + 4 h + {0..3} -- the rows ONE lane of the 32x32 accumulator layout owns, so the
// kernel's epilogue writes whole 16-byte activation units (conflict-free) and the B operand of the next stage is that unit.
// swap[s] = 1 exchanges the two slots (stages with no valid block below 8).  Row blocks outside [mb_lo, mb_hi) are zero (their loads are skipped by the kernel); `slack` zero steps follow the last stage.
size_t pack_stream_p(std::vector<uint16_t>& blob, const std::function<float(int, int)> (&A)[NSTAGE], const int (&kmax)[NSTAGE],
                     const int (&lo)[NSTAGE], const int (&hi)[NSTAGE], int (&grp)[NSTAGE], float (&unscale)[NSTAGE],
                     int (&swap)[NSTAGE], int* steps_out) {
  int total = 0;
  for (int s = 0; s < NSTAGE; ++s) { grp[s] = ((kmax[s] + 15) / 16 + 3) / 4; total += 4 * grp[s]; }
  const int slack = 4;
  const int T = total + slack;
  const size_t off = (blob.size() + 63) & ~size_t(63);
  blob.resize(off + (size_t)NWAVE * T * 2 * 64 * 8, 0);
  int t0 = 0;
  for (int s = 0; s < NSTAGE; ++s) {
    const Dense As = materialise(A[s], lo[s], hi[s], kmax[s], kmax[s]);
    float mx = 0.f;
    for (int r = lo[s] * 32; r < hi[s] * 32; ++r)
      for (int c = 0; c < kmax[s]; ++c) mx = fmaxf(mx, fabsf(As(r, c)));
    int shift = 12;
    while (shift > -12 && ldexpf(mx, shift) >= 32768.f) --shift;
    unscale[s] = ldexpf(1.f, -shift);
    swap[s] = lo[s] >= 8 ? 1 : 0;     // every valid block in the upper half: it becomes the wave's slot 0
    for (int w = 0; w < NWAVE; ++w)
      for (int t = 0; t < 4 * grp[s]; ++t)
        for (int rb = 0; rb < 2; ++rb) {
          const int mb = w + 8 * ((rb + swap[s]) & 1);
          if (mb < lo[s] || mb >= hi[s]) continue;
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const int r = mb * 32 + (lane & 31);
              const int c = 16 * t + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
              const float v = c < kmax[s] ? ldexpf(As(r, c), shift) : 0.f;
              const __half hv = __float2half_rn(v);
              blob[off + ((((size_t)w * T + t0 + t) * 2 + rb) * 64 + lane) * 8 + e] = *reinterpret_cast<const uint16_t*>(&hv);
            }
        }
    t0 += 4 * grp[s];
  }
  *steps_out = T;
  return off;
}

}  // namespace

extern "C" int hm_decoder_create(int latent_dim, const float* const* W, const float* const* bias,
                                 hm_decoder_s** out) {
  if (out == nullptr || W == nullptr || bias == nullptr) { hm_set_error("null argument"); return -1; }
  const int L = latent_dim;
  if (L < 32 || L > MAX_L || (L % 32) != 0) {
    hm_set_error("latent_dim %d unsupported: need a multiple of 32 in [32, %d]", L, MAX_L);
    return -1;
  }
  const int D0 = L + 3, m = HID - D0, m_pad = HID - L, mb_zx = 16 - L / 32;
  // row-major shapes: W0 (512, D0), W1..2 (512,512), W3 (m,512), W4..7 (512,512), W8 (1,512)
  auto w = [&](int l, int r, int c, int ld) { return W[l][(size_t)r * ld + c]; };

  Blob blob;
  std::vector<uint16_t> hblob;
  size_t hoff[NSTAGE];
  StageDescH sth[NSTAGE];
  struct Pending { size_t wp, bias; int has_bias; };
  Pending pend[NSTAGE];
  StageDesc st[NSTAGE];
  memset(st, 0, sizeof(st));
  std::function<float(int, int)> Afun[NSTAGE];      // kept for the plain-fp16 kernel's weight streams (pack_stream_p)
  int s_kmax[NSTAGE], s_lo[NSTAGE], s_hi[NSTAGE];
  auto set = [&](int s, int n_kg, int lo, int hi, int epi, int layer, int inst_bias,
                 const std::function<float(int, int)>& A, const std::function<float(int)>* bfun) {
    Afun[s] = A; s_kmax[s] = n_kg * 8; s_lo[s] = lo; s_hi[s] = hi;
    st[s].n_kg = n_kg; st[s].mb_lo = lo; st[s].mb_hi = hi; st[s].epi = epi; st[s].layer = layer;
    st[s].inst_bias = inst_bias;
    const int kmax = n_kg * 8;
    const int n_k16 = (kmax + 15) / 16;
    const Dense Ad = materialise(A, lo, hi, kmax, n_k16 * 16);     // K padded to a multiple of 16 with zero columns
    pend[s].wp = pack_stage(blob, lo, hi, n_kg, Ad);
    {   // same stage for the f16x3 kernel
      sth[s].n_k16 = n_k16;
      sth[s].mb_stride = n_k16 * 128;
      hoff[s] = pack_stage_h(hblob, lo, hi, n_k16, sth[s].mb_stride, Ad, &sth[s].unscale);
    }
    pend[s].has_bias = 0;
    if (bfun) {
      pend[s].bias = blob.alloc(HID);
      for (int f = 0; f < HID; ++f) blob.host[pend[s].bias + f] = (*bfun)(f);
      pend[s].has_bias = 1;
    }
  };
  auto plain_bias = [&](int l) { return std::function<float(int)>([=](int f) { return bias[l][f]; }); };

  // ---- forward stages (deep_sdf_decoder.py:85-105) ----
  set(0, 1, 0, 16, EPI_FWD, 0, 1, [&](int r, int c) { return c < 3 ? w(0, r, L + c, D0) : 0.f; }, nullptr);
  { auto b1 = plain_bias(1); set(1, 64, 0, 16, EPI_FWD, 1, 0, [&](int r, int c) { return w(1, r, c, HID); }, &b1); }
  { auto b2 = plain_bias(2); set(2, 64, 0, 16, EPI_FWD, 2, 0, [&](int r, int c) { return w(2, r, c, HID); }, &b2); }
  { std::function<float(int)> b3 = [&](int f) { return f < m ? bias[3][f] : 0.f; };
    set(3, 64, 0, m_pad / 32, EPI_FWD3, 3, 0, [&](int r, int c) { return r < m ? w(3, r, c, HID) : 0.f; }, &b3); }
  // lin4 input is [h3 (m) | z (L) | xyz (3)] (:87-88); z is folded into c4, xyz sits in rows m..m+2 of X
  set(4, m_pad / 8, 0, 16, EPI_FWD, 4, 2,
      [&](int r, int c) { return c < m ? w(4, r, c, HID) : (c < m + 3 ? w(4, r, c + L, HID) : 0.f); }, nullptr);
  for (int l = 5; l <= 7; ++l) {
    auto bl = plain_bias(l);
    set(l, 64, 0, 16, l == 7 ? EPI_FWD7 : EPI_FWD, l, 0, [&, l](int r, int c) { return w(l, r, c, HID); }, &bl);
  }
  // ---- backward stages: G_{l-1} = (G_l . mask_l) W_l  (utils.py:112-122 restated) ----
  for (int i = 0; i < 3; ++i) {   // transposes of lin7, lin6, lin5 -> masks of lin6, lin5, lin4
    const int l = 7 - i;
    set(8 + i, 64, 0, 16, EPI_BWD, l - 1, 0, [&, l](int r, int c) { return w(l, c, r, HID); }, nullptr);
  }
  // transpose of lin4: rows [0,m) -> d/d h3 (mask of lin3), rows [m_pad, 512) -> d/d z
  set(11, 64, 0, 16, EPI_BWD4, 3, 0,
      [&](int r, int c) { return r < m ? w(4, c, r, HID) : (r < m_pad ? 0.f : w(4, c, m + (r - m_pad), HID)); },
      nullptr);
  set(12, m_pad / 8, 0, 16, EPI_BWD, 2, 0, [&](int r, int c) { return c < m ? w(3, c, r, HID) : 0.f; }, nullptr);
  set(13, 64, 0, 16, EPI_BWD, 1, 0, [&](int r, int c) { return w(2, c, r, HID); }, nullptr);
  set(14, 64, 0, 16, EPI_BWD, 0, 0, [&](int r, int c) { return w(1, c, r, HID); }, nullptr);
  // transpose of lin0's latent columns, on the same (wave, slot) rows as lin4's latent rows
  set(15, 64, mb_zx, 16, EPI_BWD0, 0, 0, [&](int r, int c) { return w(0, c, r - m_pad, D0); }, nullptr);

  const size_t o_w8 = blob.alloc(HID), o_w0x = blob.alloc(HID * 4), o_w4x = blob.alloc(HID * 4);
  const size_t o_w0z = blob.alloc((size_t)L * HID), o_w4z = blob.alloc((size_t)L * HID);
  const size_t o_b0 = blob.alloc(HID), o_b4 = blob.alloc(HID);
  for (int f = 0; f < HID; ++f) {
    blob.host[o_w8 + f] = W[8][f];
    for (int c = 0; c < 3; ++c) {
      blob.host[o_w0x + f * 4 + c] = w(0, f, L + c, D0);
      blob.host[o_w4x + f * 4 + c] = w(4, f, m + L + c, HID);
    }
    for (int j = 0; j < L; ++j) {
      blob.host[o_w0z + (size_t)j * HID + f] = w(0, f, j, D0);
      blob.host[o_w4z + (size_t)j * HID + f] = w(4, f, m + j, HID);
    }
    blob.host[o_b0 + f] = bias[0][f];
    blob.host[o_b4 + f] = bias[4][f];
  }

  // plain-fp16 kernel: the transpose of lin4 also carries lin4's three xyz columns, as rows m..m+2 (the rows the forward
  // splice puts xyz in): d sdf / d xyz through lin4 then comes out of the matrix pipe with the rest of that stage
  Afun[11] = [&](int r, int c) {
    return r < m ? w(4, c, r, HID) : (r < m + 3 ? w(4, c, m + L + (r - m), HID) : (r < m_pad ? 0.f : w(4, c, m + (r - m_pad), HID)));
  };
  int pgrp[NSTAGE], pswap[NSTAGE], psteps = 0;
  float pus[NSTAGE];
  const size_t poff = pack_stream_p(hblob, Afun, s_kmax, s_lo, s_hi, pgrp, pus, pswap, &psteps);

  hm_decoder_s* d = new hm_decoder_s();
  d->L = L;
  d->precision = 0;
  const size_t fbytes = (blob.host.size() * sizeof(float) + 255) & ~size_t(255);
  d->blob_bytes = fbytes + hblob.size() * sizeof(uint16_t);
  hipError_t e = hipMalloc(&d->d_blob, d->blob_bytes);
  if (e != hipSuccess) { hm_set_error("hipMalloc(%zu) failed: %s", d->blob_bytes, hipGetErrorString(e)); delete d; return -2; }
  e = hipMemcpy(d->d_blob, blob.host.data(), blob.host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(static_cast<char*>(d->d_blob) + fbytes, hblob.data(), hblob.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) { hm_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); (void)hipFree(d->d_blob); delete d; return -2; }
  const float* base = static_cast<const float*>(d->d_blob);
  DecoderDev& dv = d->dev;
  dv.L = L; dv.m = m; dv.m_pad = m_pad; dv.mb_zx = mb_zx;
  dv.ps = static_cast<const char*>(d->d_blob) + fbytes + poff * sizeof(uint16_t);
  dv.ps_steps = psteps;
  for (int s = 0; s < NSTAGE; ++s) { dv.pgrp[s] = pgrp[s]; dv.pus[s] = pus[s]; dv.pswap[s] = pswap[s]; }
  for (int s = 0; s < NSTAGE; ++s) {
    dv.st[s] = st[s];
    dv.st[s].wp = base + pend[s].wp;
    dv.st[s].bias = pend[s].has_bias ? base + pend[s].bias : nullptr;
    dv.sth[s] = sth[s];
    dv.sth[s].wp = static_cast<const char*>(d->d_blob) + fbytes + hoff[s] * sizeof(uint16_t);
  }
  dv.w8 = base + o_w8; dv.b8 = bias[8][0];
  dv.w0x = base + o_w0x; dv.w4x = base + o_w4x; dv.w0z = base + o_w0z; dv.w4z = base + o_w4z;
  dv.b0 = base + o_b0; dv.b4 = base + o_b4;
  *out = d;
  return 0;
}

// Layer table of an arbitrary reference `Decoder` (deep_sdf_decoder.py:29-72), mirrored from include/hortimapping_amd.h
struct hm_decoder_arch {
  int latent_dim, n_lin, use_tanh;
  int in_dim[HM_ANY_MAX_LIN], out_dim[HM_ANY_MAX_LIN], cat[HM_ANY_MAX_LIN], layer_norm[HM_ANY_MAX_LIN];
};

extern "C" int hm_decoder_create_arch(const hm_decoder_arch* arch, const float* const* W, const float* const* bias,
                                      const float* const* ln_weight, const float* const* ln_bias, hm_decoder_s** out) {
  if (out == nullptr || arch == nullptr || W == nullptr || bias == nullptr) { hm_set_error("null argument"); return -1; }
  const int L = arch->latent_dim, n = arch->n_lin, D0 = L + 3;
  if (L < 32 || L > MAX_L || (L % 32) != 0) {
    hm_set_error("latent_dim %d unsupported: need a multiple of 32 in [32, %d]", L, MAX_L); return -1; }
  if (n < 2 || n > HM_ANY_MAX_LIN) { hm_set_error("n_lin %d unsupported: need 2 .. %d Linear layers", n, HM_ANY_MAX_LIN); return -1; }
  int n_ln = 0;
  for (int l = 0; l < n; ++l) {
    const int in = arch->in_dim[l], od = arch->out_dim[l], cat = arch->cat[l];
    if (W[l] == nullptr || bias[l] == nullptr) { hm_set_error("lin%d: null weight / bias", l); return -1; }
    if (in < 1 || in > HM_ANY_MAX_WIDTH || od < 1 || od > HM_ANY_MAX_WIDTH) {
      hm_set_error("lin%d: %d -> %d outside the supported widths 1 .. %d", l, in, od, HM_ANY_MAX_WIDTH); return -1; }
    if (cat < 0 || cat > 2 || (l == 0 && cat != 0)) { hm_set_error("lin%d: bad concatenation code %d", l, cat); return -1; }
    const int expect = l == 0 ? D0 : arch->out_dim[l - 1] + (cat == 1 ? D0 : (cat == 2 ? 3 : 0));
    if (in != expect) {   // deep_sdf_decoder.py:41-47: out_dim of the previous layer leaves room for the concatenation
      hm_set_error("lin%d: in_dim %d does not match the previous layer's output (+ concatenation) %d", l, in, expect); return -1; }
    if (arch->layer_norm[l]) {
      if (l == n - 1) { hm_set_error("lin%d: LayerNorm on the last layer is never applied by Decoder.forward", l); return -1; }
      if (ln_weight == nullptr || ln_bias == nullptr || ln_weight[l] == nullptr || ln_bias[l] == nullptr) {
        hm_set_error("lin%d: LayerNorm parameters missing", l); return -1; }
      ++n_ln;
    }
  }
  if (arch->out_dim[n - 1] != 1) { hm_set_error("the last layer must have one output (sdf)"); return -1; }

  Blob blob;
  std::vector<uint16_t> hblob;
  size_t o_wf[HM_ANY_MAX_LIN], o_wb[HM_ANY_MAX_LIN], o_b[HM_ANY_MAX_LIN], o_g[HM_ANY_MAX_LIN], o_be[HM_ANY_MAX_LIN];
  size_t h_wf[HM_ANY_MAX_LIN], h_wb[HM_ANY_MAX_LIN];
  float usf[HM_ANY_MAX_LIN], usb[HM_ANY_MAX_LIN];
  for (int l = 0; l < n; ++l) {
    const int in = arch->in_dim[l], od = arch->out_dim[l];
    const float* Wl = W[l];
    const std::function<float(int, int)> Af = [&](int r, int c) { return (r < od && c < in) ? Wl[(size_t)r * in + c] : 0.f; };
    const std::function<float(int, int)> Ab = [&](int r, int c) { return (r < in && c < od) ? Wl[(size_t)c * in + r] : 0.f; };
    const Dense Afd = materialise(Af, 0, (od + 31) / 32, ((in + 15) / 16) * 16, ((in + 15) / 16) * 16);
    const Dense Abd = materialise(Ab, 0, (in + 31) / 32, ((od + 15) / 16) * 16, ((od + 15) / 16) * 16);
    o_wf[l] = pack_stage(blob, 0, (od + 31) / 32, (in + 7) / 8, Afd);
    o_wb[l] = pack_stage(blob, 0, (in + 31) / 32, (od + 7) / 8, Abd);
    h_wf[l] = pack_stage_h(hblob, 0, (od + 31) / 32, (in + 15) / 16, ((in + 15) / 16) * 128, Afd, &usf[l]);
    h_wb[l] = pack_stage_h(hblob, 0, (in + 31) / 32, (od + 15) / 16, ((od + 15) / 16) * 128, Abd, &usb[l]);
    o_b[l] = blob.alloc(HM_ANY_MAX_WIDTH);
    for (int f = 0; f < od; ++f) blob.host[o_b[l] + f] = bias[l][f];
    o_g[l] = o_be[l] = 0;
    if (arch->layer_norm[l]) {
      o_g[l] = blob.alloc(HM_ANY_MAX_WIDTH);
      o_be[l] = blob.alloc(HM_ANY_MAX_WIDTH);
      for (int f = 0; f < od; ++f) { blob.host[o_g[l] + f] = ln_weight[l][f]; blob.host[o_be[l] + f] = ln_bias[l][f]; }
    }
  }
  hm_decoder_s* d = new hm_decoder_s();
  d->L = L;
  d->precision = 0;
  d->generic = 1;
  const size_t fbytes = (blob.host.size() * sizeof(float) + 255) & ~size_t(255);
  d->blob_bytes = fbytes + hblob.size() * sizeof(uint16_t);
  hipError_t e = hipMalloc(&d->d_blob, d->blob_bytes);
  if (e != hipSuccess) { hm_set_error("hipMalloc(%zu) failed: %s", d->blob_bytes, hipGetErrorString(e)); delete d; return -2; }
  e = hipMemcpy(d->d_blob, blob.host.data(), blob.host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(static_cast<char*>(d->d_blob) + fbytes, hblob.data(), hblob.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hm_set_error("device allocation / copy failed: %s", hipGetErrorString(e));
    (void)hipFree(d->d_blob); delete d; return -2;
  }
  const float* base = static_cast<const float*>(d->d_blob);
  AnyDev& av = d->any;
  av.L = L; av.n_lin = n; av.use_tanh = arch->use_tanh ? 1 : 0; av.n_ln = n_ln;
  for (int l = 0; l < n; ++l) {
    AnyLayer& ly = av.lay[l];
    ly.wf = base + o_wf[l]; ly.wb = base + o_wb[l]; ly.bias = base + o_b[l];
    ly.gamma = arch->layer_norm[l] ? base + o_g[l] : nullptr;
    ly.beta = arch->layer_norm[l] ? base + o_be[l] : nullptr;
    const char* hbase = static_cast<const char*>(d->d_blob) + fbytes;
    ly.wfh = hbase + h_wf[l] * sizeof(uint16_t); ly.wbh = hbase + h_wb[l] * sizeof(uint16_t);
    ly.kf16 = (arch->in_dim[l] + 15) / 16; ly.kb16 = (arch->out_dim[l] + 15) / 16;
    ly.usf = usf[l]; ly.usb = usb[l];
    ly.in_dim = arch->in_dim[l]; ly.out_dim = arch->out_dim[l]; ly.cat = arch->cat[l];
    ly.ln = arch->layer_norm[l] ? 1 : 0;
  }
  *out = d;
  return 0;
}

extern "C" int hm_decoder_destroy(hm_decoder_s* d) {
  if (d == nullptr) return 0;
  (void)hipFree(d->d_blob);
  delete d;
  return 0;
}

extern "C" int hm_decoder_latent_dim(const hm_decoder_s* d) { return d ? d->L : -1; }

extern "C" int hm_decoder_set_precision(hm_decoder_s* d, int precision) {
  if (d == nullptr) { hm_set_error("null decoder"); return -1; }
  if (d->generic && precision != 0 && precision != 1) {
    hm_set_error("a decoder built by hm_decoder_create_arch runs in exact fp32 (precision 0) or f16x3 (precision 1) only"); return -1; }
  if (precision < 0 || precision > 3) { hm_set_error("precision must be 0 (f32), 1 (f16x3), 2 (f16x3f_f16b) or 3 (f16)"); return -1; }
  d->precision = precision;
  return 0;
}

extern "C" int hm_decoder_get_precision(const hm_decoder_s* d) { return d ? d->precision : -1; }
