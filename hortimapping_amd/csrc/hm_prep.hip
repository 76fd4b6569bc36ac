// K7: per-instance render-data preparation on the device -- the image scans of `get_render_data`
// (reference wild_completion/utils.py:39-109): for every (fruit instance, frame) pair the count and bounding box of the
// instance's valid mask pixels (:52-63), then, inside the padded box in raster order, the foreground candidates
// (mask and depth > 0, :85-88) and the background candidates (not mask, :74-77), of which the reference keeps a random
// subset drawn by np.random.choice (:78-82, :89-93).  The draw itself stays on the host (MT19937 under the caller's
// global seed is sequential and must be consumed in the reference's order): pass 2 returns the candidate counts, the
// host draws ranks, pass 3 gathers the chosen candidates by rank and writes pixels, depths and ray directions
// K^-1 [u, v, 1] (get_rays, utils.py:23-37).  HBM-bound integer scans: one read of the id + depth images for the
// statistics, one read of each box per scan pass; no LDS tiling needed, wave ballots do the ordered compaction.
#include <limits.h>

#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

namespace {

constexpr int NTP = 256;

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}

// stats[slot][f][5] = {count, v_min, v_max, u_min, u_max} of pixels with lut[id] == slot and depth > 0.
// Grid (x: pixel blocks, y: frame).  Atomics are aggregated per wave and instance: a wave usually sees 0-2 instances.
__global__ __launch_bounds__(NTP) void k_prep_stats(const int* __restrict__ id_imgs, const float* __restrict__ depth,
                                                    int HW, int W, int F, const int* __restrict__ lut, int lut_size,
                                                    int* __restrict__ stats) {
  const int f = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int* idf = id_imgs + (size_t)f * HW;
  const float* df = depth + (size_t)f * HW;
  for (int base = (blockIdx.x * NTP + (threadIdx.x & ~63)); base < HW; base += gridDim.x * NTP) {   // wave-uniform trip count
    const int p = base + lane;
    int slot = -1, v = 0, u = 0;
    if (p < HW) {
      const int idv = idf[p];
      if (idv >= 0 && idv < lut_size) slot = lut[idv];
      if (slot >= 0 && !(df[p] > 0.f)) slot = -1;
      v = p / W;
      u = p - v * W;
    }
    unsigned long long act = __ballot(slot >= 0);
    while (act) {
      const int leader = __ffsll((long long)act) - 1;
      const int s = __shfl(slot, leader);
      const bool mine = slot == s;
      const unsigned long long same = __ballot(mine);
      const int vmin = wave_min(mine ? v : INT_MAX), vmax = wave_max(mine ? v : -1);
      const int umin = wave_min(mine ? u : INT_MAX), umax = wave_max(mine ? u : -1);
      if (lane == leader) {
        int* st = stats + ((size_t)s * F + f) * 5;
        atomicAdd(st + 0, __popcll(same));
        atomicMin(st + 1, vmin);
        atomicMax(st + 2, vmax);
        atomicMin(st + 3, umin);
        atomicMax(st + 4, umax);
      }
      act &= ~same;
    }
  }
}

// One workgroup per (instance, frame) pair: ordered scan of the padded box.  pairs[p] = {sid, f, min_v, max_v, min_u,
// max_u, n_sel_bg, n_sel_fg}; n_sel < 0: keep every candidate (there are at most `cap`).
// GATHER = false: counts[p] = {n_bg, n_fg}.  GATHER = true: candidate of rank r is written to output row perm[i] when
// sel[i] == r (sel sorted ascending), or to row r when every candidate is kept.
template <bool GATHER>
__global__ __launch_bounds__(NTP) void k_prep_scan(const int* __restrict__ id_imgs, const float* __restrict__ depth,
                                                   int H, int W, const int* __restrict__ pairs, int* __restrict__ counts,
                                                   const int* __restrict__ sel, const int* __restrict__ perm, int cap,
                                                   const double* __restrict__ invK, int* __restrict__ pix,
                                                   float* __restrict__ dep, float* __restrict__ rays) {
  extern __shared__ int smem[];            // GATHER: sel[2][cap] | perm[2][cap]; then wave totals [4][2]
  const int p = blockIdx.x;
  const int* pr = pairs + (size_t)p * 8;
  const int sid = pr[0], f = pr[1], min_v = pr[2], max_v = pr[3], min_u = pr[4], max_u = pr[5];
  const int n_sel[2] = {pr[6], pr[7]};
  const int bw = max_u - min_u + 1, bh = max_v - min_v + 1;
  const int n = bw * bh;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int* s_sel = smem;
  int* s_perm = smem + (GATHER ? 2 * cap : 0);
  int* s_tot = smem + (GATHER ? 4 * cap : 0);
  if (GATHER) {
    for (int i = tid; i < 2 * cap; i += NTP) {
      s_sel[i] = sel[(size_t)p * 2 * cap + i];
      s_perm[i] = perm[(size_t)p * 2 * cap + i];
    }
  }
  double k[9];
  if (GATHER) {
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = invK[i];
  }
  const int* idf = id_imgs + (size_t)f * H * W;
  const float* df = depth + (size_t)f * H * W;
  int run[2] = {0, 0};
  __syncthreads();
  for (int base = 0; base < n; base += NTP) {
    const int idx = base + tid;
    bool flag[2] = {false, false};
    int v = 0, u = 0;
    float d = 0.f;
    if (idx < n) {
      const int r = idx / bw;
      v = min_v + r;
      u = min_u + (idx - r * bw);
      const int idv = idf[(size_t)v * W + u];
      d = df[(size_t)v * W + u];
      flag[0] = idv != sid;                       // background: ~mask_bool                (utils.py:74)
      flag[1] = idv == sid && d > 0.f;            // foreground: mask_bool & (depth > 0)   (:51, :85)
    }
    int pre[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned long long m = __ballot(flag[c]);
      pre[c] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_tot[wv * 2 + c] = __popcll(m);
    }
    __syncthreads();
    int tot[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int off = 0;
#pragma unroll
      for (int w2 = 0; w2 < NTP / 64; ++w2) {
        const int t = s_tot[w2 * 2 + c];
        if (w2 < wv) off += t;
        tot[c] += t;
      }
      if (GATHER && flag[c]) {
        const int rank = run[c] + off + pre[c];
        int row = -1;
        if (n_sel[c] < 0) {
          row = rank < cap ? rank : -1;
        } else {
          int lo = 0, hi = n_sel[c] - 1;
          const int* ss = s_sel + c * cap;
          while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const int sv = ss[mid];
            if (sv == rank) { row = s_perm[c * cap + mid]; break; }
            if (sv < rank) lo = mid + 1; else hi = mid - 1;
          }
        }
        if (row >= 0) {
          const size_t o = ((size_t)p * 2 + c) * cap + row;
          pix[o * 2 + 0] = u;
          pix[o * 2 + 1] = v;
          dep[o] = d;
          const double ud = (double)u, vd = (double)v;
#pragma unroll
          for (int r = 0; r < 3; ++r)      // (u k0 + v k1) + 1 k2 in fp64, products and sums rounded separately (numpy)
            rays[o * 3 + r] = (float)__dadd_rn(__dadd_rn(__dmul_rn(ud, k[3 * r]), __dmul_rn(vd, k[3 * r + 1])), k[3 * r + 2]);
        }
      }
      run[c] += tot[c];
    }
    __syncthreads();
  }
  if (!GATHER && tid == 0) {
    counts[(size_t)p * 2 + 0] = run[0];
    counts[(size_t)p * 2 + 1] = run[1];
  }
}


// ---- clean_pcd (utils.py:407-417): DBSCAN of one instance's surface samples, one workgroup per instance.
// Points (fp64, as the KD-tree of the reference's / scikit-learn's DBSCAN sees them) and component ids live in LDS; every
// pass is a brute-force n x n sweep (n <= DB_MAXN: 26 M distance tests for 5,120 points, microseconds on one CU).
//   1. neighbour counts (self included, d <= eps): core points have >= min_pts;
//   2. connected components of the core points under eps-adjacency by min-label propagation with pointer jumping
//      (label = smallest core index of the component; the fixed point is unique, so the result is deterministic);
//   3. a non-core point with core neighbours joins the adjacent component with the SMALLEST first core index -- the
//      cluster scikit-learn's index-ordered expansion reaches it from first; without core neighbours it is noise (-1).
// Output comp[b][i] = smallest core index of i's cluster, or -1: the host turns ranks of these into DBSCAN labels.
constexpr int DB_NT = 1024;
constexpr int DB_MAXN = 5120;

__global__ __launch_bounds__(DB_NT) void k_dbscan(const double* __restrict__ pts, const int* __restrict__ n_pts,
                                                  int n_stride, double eps, const int* __restrict__ min_pts,
                                                  int* __restrict__ comp_out) {
  extern __shared__ double dsm[];
  double* px = dsm;                       // [3][DB_MAXN]
  int* comp = reinterpret_cast<int*>(dsm + 3 * DB_MAXN);
  __shared__ int changed;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_pts[b];
  const int mp = min_pts[b];
  const double e2 = eps * eps;
  const double* pb = pts + (size_t)b * n_stride * 3;
  for (int i = tid; i < n; i += DB_NT) {
    px[i] = pb[3 * i];
    px[DB_MAXN + i] = pb[3 * i + 1];
    px[2 * DB_MAXN + i] = pb[3 * i + 2];
  }
  __syncthreads();
  constexpr int PER = DB_MAXN / DB_NT;
  for (int r = 0; r < PER; ++r) {
    const int i = tid + r * DB_NT;
    if (i < n) {
      const double x = px[i], y = px[DB_MAXN + i], z = px[2 * DB_MAXN + i];
      int cnt = 0;
      for (int j = 0; j < n; ++j) {
        const double dx = px[j] - x, dy = px[DB_MAXN + j] - y, dz = px[2 * DB_MAXN + j] - z;
        cnt += (dx * dx + dy * dy + dz * dz <= e2) ? 1 : 0;
      }
      comp[i] = cnt >= mp ? i : INT_MAX;
    }
  }
  __syncthreads();
  for (int it = 0; it < 4 * DB_MAXN; ++it) {
    if (tid == 0) changed = 0;
    __syncthreads();
    for (int r = 0; r < PER; ++r) {
      const int i = tid + r * DB_NT;
      if (i < n && comp[i] != INT_MAX) {
        const double x = px[i], y = px[DB_MAXN + i], z = px[2 * DB_MAXN + i];
        int m = comp[i];
        for (int j = 0; j < n; ++j) {
          const int cj = comp[j];
          if (cj < m) {
            const double dx = px[j] - x, dy = px[DB_MAXN + j] - y, dz = px[2 * DB_MAXN + j] - z;
            if (dx * dx + dy * dy + dz * dz <= e2) m = cj;
          }
        }
        const int mm = comp[m];               // pointer jump: the label of my label (labels are core indices)
        if (mm < m) m = mm;
        if (m < comp[i]) { comp[i] = m; changed = 1; }
      }
    }
    __syncthreads();
    if (!changed) break;
    __syncthreads();
  }
  int* out = comp_out + (size_t)b * n_stride;
  for (int r = 0; r < PER; ++r) {
    const int i = tid + r * DB_NT;
    if (i < n) {
      int m = comp[i];
      if (m == INT_MAX) {                      // border or noise
        const double x = px[i], y = px[DB_MAXN + i], z = px[2 * DB_MAXN + i];
        for (int j = 0; j < n; ++j) {
          const int cj = comp[j];
          if (cj < m) {
            const double dx = px[j] - x, dy = px[DB_MAXN + j] - y, dz = px[2 * DB_MAXN + j] - z;
            if (dx * dx + dy * dy + dz * dz <= e2) m = cj;
          }
        }
      }
      out[i] = m == INT_MAX ? -1 : m;
    }
  }
}

// ---- get_pose_init's crop of the background cloud (utils.py:442-447): indices, in input order, of the points inside an
// axis-aligned box (closed bounds, fp64 compares like numpy); one workgroup per box.  GATHER false: counts only.
template <bool GATHER>
__global__ __launch_bounds__(NTP) void k_box_select(const double* __restrict__ pts, int n, const double* __restrict__ boxes,
                                                    int* __restrict__ counts, const long long* __restrict__ offsets,
                                                    int* __restrict__ idx_out) {
  __shared__ int s_tot[NTP / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double lo[3], hi[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { lo[c] = boxes[b * 6 + c]; hi[c] = boxes[b * 6 + 3 + c]; }
  int run = 0;
  int* out = GATHER ? idx_out + offsets[b] : nullptr;
  for (int base = 0; base < n; base += NTP) {
    const int i = base + tid;
    bool in = false;
    if (i < n) {
      const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
      in = x >= lo[0] && x <= hi[0] && y >= lo[1] && y <= hi[1] && z >= lo[2] && z <= hi[2];
    }
    const unsigned long long m = __ballot(in);
    const int pre = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_tot[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < NTP / 64; ++w2) {
      if (w2 < wv) off += s_tot[w2];
      tot += s_tot[w2];
    }
    if (GATHER && in) out[run + off + pre] = i;
    run += tot;
    __syncthreads();
  }
  if (!GATHER && tid == 0) counts[b] = run;
}

}  // namespace

extern "C" int hm_prep_stats(const int* d_id_imgs, const float* d_depth, int F, int H, int W, const int* d_lut,
                             int lut_size, int B, int* d_stats, void* stream) {
  if (F < 0 || H <= 0 || W <= 0 || B < 0 || lut_size < 0) { hm_set_error("hm_prep_stats: bad argument"); return -1; }
  if (F == 0 || B == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int HW = H * W;
  int gx = (HW + NTP - 1) / NTP;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_prep_stats, dim3(gx, F), dim3(NTP), 0, st, d_id_imgs, d_depth, HW, W, F, d_lut, lut_size, d_stats);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int hm_prep_scan(const int* d_id_imgs, const float* d_depth, int H, int W, const int* d_pairs, int P,
                            int gather, int* d_counts, const int* d_sel, const int* d_perm, int cap,
                            const double* d_invK, int* d_pix, float* d_depth_out, float* d_rays, void* stream) {
  if (H <= 0 || W <= 0 || P < 0 || cap < 0 || cap > 8192) { hm_set_error("hm_prep_scan: bad argument"); return -1; }
  if (P == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gather) {
    const size_t sm = ((size_t)4 * cap + 8) * sizeof(int);         // up to 128 KiB at cap 8192: beyond the 64 KiB default
    HM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prep_scan<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipLaunchKernelGGL((k_prep_scan<true>), dim3(P), dim3(NTP), sm, st, d_id_imgs, d_depth, H, W, d_pairs, d_counts,
                       d_sel, d_perm, cap, d_invK, d_pix, d_depth_out, d_rays);
  } else {
    hipLaunchKernelGGL((k_prep_scan<false>), dim3(P), dim3(NTP), 8 * sizeof(int), st, d_id_imgs, d_depth, H, W, d_pairs,
                       d_counts, d_sel, d_perm, cap, d_invK, d_pix, d_depth_out, d_rays);
  }
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int hm_prep_dbscan(const double* d_pts, const int* d_n_pts, int n_stride, int B, double eps, const int* d_min_pts,
                              int* d_comp, void* stream) {
  if (B < 0 || n_stride <= 0 || n_stride > DB_MAXN || !(eps > 0.0)) {
    hm_set_error("hm_prep_dbscan: bad argument (at most %d points per instance)", DB_MAXN);
    return -1;
  }
  if (B == 0) return 0;
  const size_t sm = (size_t)3 * DB_MAXN * sizeof(double) + (size_t)DB_MAXN * sizeof(int);
  // function attributes are per device: set on every call (a host-side table write) rather than cached per process
  HM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dbscan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
  hipLaunchKernelGGL(k_dbscan, dim3(B), dim3(DB_NT), sm, static_cast<hipStream_t>(stream), d_pts, d_n_pts, n_stride, eps,
                     d_min_pts, d_comp);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int hm_prep_box_select(const double* d_pts, int n, const double* d_boxes, int B, int gather, int* d_counts,
                                  const long long* d_offsets, int* d_idx, void* stream) {
  if (n < 0 || B < 0) { hm_set_error("hm_prep_box_select: bad argument"); return -1; }
  if (B == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gather) hipLaunchKernelGGL((k_box_select<true>), dim3(B), dim3(NTP), 0, st, d_pts, n, d_boxes, d_counts, d_offsets, d_idx);
  else hipLaunchKernelGGL((k_box_select<false>), dim3(B), dim3(NTP), 0, st, d_pts, n, d_boxes, d_counts, d_offsets, d_idx);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}
