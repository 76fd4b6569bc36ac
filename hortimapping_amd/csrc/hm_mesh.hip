// Iso-surface extraction from the decoded SDF grid (SURVEY.md 8f "next" row 1).
//
// The reference decodes a voxels_dim^3 grid (wild_completion/mesher.py:14-24, utils.py:542-562) and runs
// scikit-image's marching cubes on the host (utils.py:565-588).  scikit-image is not part of this image, so the
// surface is extracted on the GPU with marching TETRAHEDRA over the same grid: every cell is split into the six
// Kuhn tetrahedra around its main diagonal (translation invariant, so neighbouring cells agree on shared faces and the
// surface is watertight), each tetrahedron emits 0, 1 or 2 triangles with vertices linearly interpolated along grid
// edges exactly as marching cubes interpolates them.  Output is a triangle soup per instance in object coordinates
// (grid spans [-1, 1]^3 * cube_radius, vertex = (-1 + index * 2/(n-1)) * cube_radius, utils.py:577-586); edge
// vertices are evaluated from the lower to the higher grid index so that shared vertices are bit-identical and the
// host can weld them with an exact `unique`.  Two passes (count, scan, emit) keep the triangle order deterministic.
// HBM-bound integer/byte work: one thread per cell, coalesced along z.
#include <mutex>

#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

namespace {

__constant__ int c_tet[6][4] = {{0, 5, 1, 6}, {0, 1, 2, 6}, {0, 2, 3, 6}, {0, 3, 7, 6}, {0, 7, 4, 6}, {0, 4, 5, 6}};
__constant__ int c_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

struct Corner { float s; int ix, iy, iz; int gid; };

__device__ __forceinline__ void interp(const Corner& a, const Corner& b, float h, float radius, float level,
                                       float* out) {
  // always from the lower to the higher global grid index: bit-identical for every cell that shares the edge
  const Corner& lo = a.gid < b.gid ? a : b;
  const Corner& hi = a.gid < b.gid ? b : a;
  const float t = (level - lo.s) / (hi.s - lo.s);
  out[0] = (-1.f + (lo.ix + t * (hi.ix - lo.ix)) * h) * radius;
  out[1] = (-1.f + (lo.iy + t * (hi.iy - lo.iy)) * h) * radius;
  out[2] = (-1.f + (lo.iz + t * (hi.iz - lo.iz)) * h) * radius;
}

__device__ __forceinline__ void emit_tri(float* dst, const float* p0, const float* p1, const float* p2,
                                         const float* outward) {
  // orient so that the normal points to increasing sdf (outside)
  const float ux = p1[0] - p0[0], uy = p1[1] - p0[1], uz = p1[2] - p0[2];
  const float vx = p2[0] - p0[0], vy = p2[1] - p0[1], vz = p2[2] - p0[2];
  const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  const bool flip = nx * outward[0] + ny * outward[1] + nz * outward[2] < 0.f;
  const float* q1 = flip ? p2 : p1;
  const float* q2 = flip ? p1 : p2;
  for (int i = 0; i < 3; ++i) { dst[i] = p0[i]; dst[3 + i] = q1[i]; dst[6 + i] = q2[i]; }
}

// count (dst == nullptr) or emit the triangles of one cell; returns the number of triangles
__device__ int cell_triangles(const float* __restrict__ sdf, int n, int cx, int cy, int cz, float level,
                              float radius, float* dst) {
  const float h = 2.f / (float)(n - 1);
  Corner c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k].ix = cx + c_corner[k][0]; c[k].iy = cy + c_corner[k][1]; c[k].iz = cz + c_corner[k][2];
    c[k].gid = (c[k].ix * n + c[k].iy) * n + c[k].iz;
    c[k].s = sdf[c[k].gid];
  }
  int count = 0;
  for (int t = 0; t < 6; ++t) {
    const Corner* v[4] = {&c[c_tet[t][0]], &c[c_tet[t][1]], &c[c_tet[t][2]], &c[c_tet[t][3]]};
    int in[4], nin = 0;
    for (int k = 0; k < 4; ++k) { in[k] = v[k]->s < level; nin += in[k]; }
    if (nin == 0 || nin == 4) continue;
    // reorder: inside vertices first
    const Corner* a[4];
    int ia = 0, ib = nin;
    for (int k = 0; k < 4; ++k) { if (in[k]) a[ia++] = v[k]; else a[ib++] = v[k]; }
    // outward direction: from the inside centroid to the outside centroid
    float cin[3] = {0, 0, 0}, cout[3] = {0, 0, 0};
    for (int k = 0; k < 4; ++k) {
      float* acc = k < nin ? cin : cout;
      acc[0] += a[k]->ix; acc[1] += a[k]->iy; acc[2] += a[k]->iz;
    }
    const float outward[3] = {cout[0] / (4 - nin) - cin[0] / nin, cout[1] / (4 - nin) - cin[1] / nin,
                              cout[2] / (4 - nin) - cin[2] / nin};
    if (nin == 1 || nin == 3) {
      // one vertex on one side: triangle on the three edges leaving it
      const Corner* apex = nin == 1 ? a[0] : a[3];
      const Corner* o0 = nin == 1 ? a[1] : a[0];
      const Corner* o1 = nin == 1 ? a[2] : a[1];
      const Corner* o2 = nin == 1 ? a[3] : a[2];
      if (dst != nullptr) {
        float p0[3], p1[3], p2[3];
        interp(*apex, *o0, h, radius, level, p0);
        interp(*apex, *o1, h, radius, level, p1);
        interp(*apex, *o2, h, radius, level, p2);
        emit_tri(dst + 9 * count, p0, p1, p2, outward);
      }
      count += 1;
    } else {
      // two inside (a0, a1), two outside (a2, a3): quad a0a2, a0a3, a1a3, a1a2
      if (dst != nullptr) {
        float p0[3], p1[3], p2[3], p3[3];
        interp(*a[0], *a[2], h, radius, level, p0);
        interp(*a[0], *a[3], h, radius, level, p1);
        interp(*a[1], *a[3], h, radius, level, p2);
        interp(*a[1], *a[2], h, radius, level, p3);
        emit_tri(dst + 9 * count, p0, p1, p2, outward);
        emit_tri(dst + 9 * (count + 1), p0, p2, p3, outward);
      }
      count += 2;
    }
  }
  return count;
}

// ---- marching CUBES over the same grid (the algorithm family of the reference's scikit-image call, utils.py:573):
// every vertex lies on a grid edge at the linear-interpolation crossing, so the vertex SET is the one any marching-cubes
// variant produces on this grid (oracle/level_set.py); the per-configuration triangulation is generated at load time
// (build_mc_table below) with ONE rule for ambiguous faces that depends on the face's corner signs only, so neighbouring
// cells agree on every shared face and the surface is closed.
constexpr int MC_MAXT = 8;
__constant__ unsigned char c_mc_ntri[256];
__constant__ unsigned char c_mc_tri[256][MC_MAXT * 3];       // cube-edge indices, triangles oriented outward
__constant__ int c_mc_edge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

__device__ int cell_triangles_mc(const float* __restrict__ sdf, int n, int cx, int cy, int cz, float level, float radius,
                                 float* dst) {
  const float h = 2.f / (float)(n - 1);
  Corner c[8];
  int cfg = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k].ix = cx + c_corner[k][0]; c[k].iy = cy + c_corner[k][1]; c[k].iz = cz + c_corner[k][2];
    c[k].gid = (c[k].ix * n + c[k].iy) * n + c[k].iz;
    c[k].s = sdf[c[k].gid];
    cfg |= (c[k].s < level ? 1 : 0) << k;
  }
  const int nt = c_mc_ntri[cfg];
  if (dst != nullptr) {
    for (int t = 0; t < nt; ++t) {
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int e = c_mc_tri[cfg][3 * t + v];
        interp(c[c_mc_edge[e][0]], c[c_mc_edge[e][1]], h, radius, level, dst + 9 * t + 3 * v);
      }
    }
  }
  return nt;
}

template <bool MC>
__device__ __forceinline__ int cell_tris(const float* __restrict__ sdf, int n, int cx, int cy, int cz, float level,
                                         float radius, float* dst) {
  return MC ? cell_triangles_mc(sdf, n, cx, cy, cz, level, radius, dst) : cell_triangles(sdf, n, cx, cy, cz, level, radius, dst);
}

template <bool MC>
__global__ void k_mt_count(const float* __restrict__ sdf, int n, float level, int* __restrict__ counts) {
  const int b = blockIdx.y;
  const int nc = n - 1;
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= nc * nc * nc) return;
  const int cz = cell % nc, cy = (cell / nc) % nc, cx = cell / (nc * nc);
  counts[(size_t)b * nc * nc * nc + cell] =
      cell_tris<MC>(sdf + (size_t)b * n * n * n, n, cx, cy, cz, level, 1.f, nullptr);
}

// exclusive scan of the per-cell counts, one workgroup per instance (in place), total -> tri_count[b]
__global__ __launch_bounds__(1024) void k_mt_scan(int* __restrict__ counts, int ncell, int* __restrict__ tri_count) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = blockIdx.x;
  int* c = counts + (size_t)b * ncell;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ncell; base += 1024) {
    const int i = base + tid;
    const int v = i < ncell ? c[i] : 0;
    int s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(s, off);
      if (lane >= off) s += u;
    }
    if (lane == 63) wsum[wv] = s;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < wv; ++w) pre += wsum[w];
    if (i < ncell) c[i] = pre + s - v;
    __syncthreads();
    if (tid == 1023) carry = pre + s;
    __syncthreads();
  }
  if (tid == 0) tri_count[b] = carry;
}

template <bool MC>
__global__ void k_mt_emit(const float* __restrict__ sdf, int n, float level, float radius,
                          const int* __restrict__ offsets, float* __restrict__ tris, int max_tris) {
  const int b = blockIdx.y;
  const int nc = n - 1;
  const int ncell = nc * nc * nc;
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  const int cz = cell % nc, cy = (cell / nc) % nc, cx = cell / (nc * nc);
  const int off = offsets[(size_t)b * ncell + cell];
  if (off + 12 > max_tris) {
    // capacity guard: only emit when the worst case of this cell fits (the count is still reported in full)
    const int cnt = cell_tris<MC>(sdf + (size_t)b * n * n * n, n, cx, cy, cz, level, radius, nullptr);
    if (off + cnt > max_tris) return;
  }
  cell_tris<MC>(sdf + (size_t)b * n * n * n, n, cx, cy, cz, level, radius, tris + ((size_t)b * max_tris + off) * 9);
}

}  // namespace

// Marching-cubes triangle table, generated instead of transcribed.  For each of the 256 corner-sign configurations:
// on every cube face the crossed edges are joined into segments (two crossed edges: one segment; four -- the ambiguous
// face with diagonal corners inside -- each INSIDE corner is cut off on its own, a rule that depends on that face's
// signs only); the segments close into loops over the crossed edges, each loop is triangulated by constrained ear clipping (mc_triangulate) and oriented so that
// its normal points from the inside corners to the outside ones.
// Ear clipping of one loop (cyclic list of cube-edge ids) under the rule that no triangle side other than the loop's
// own sides may join two cube edges of the same cube face: such a diagonal would lie in that face and coincide with a
// segment of another polygon of this cell or of the neighbouring one (a non-manifold edge).  A valid triangulation
// exists for every loop of every configuration (checked exhaustively; tests/test_gpu_mesher.py on random grids).
static bool mc_share_face(int a, int b) {
  static const int fedge[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 9, 4, 8}, {2, 10, 6, 11}, {3, 11, 7, 8}, {1, 10, 5, 9}};
  for (int f = 0; f < 6; ++f) {
    bool ha = false, hb = false;
    for (int k = 0; k < 4; ++k) { ha |= fedge[f][k] == a; hb |= fedge[f][k] == b; }
    if (ha && hb) return true;
  }
  return false;
}
static int mc_triangulate(const int* loop, int n, int (*out)[3]) {
  if (n == 3) { out[0][0] = loop[0]; out[0][1] = loop[1]; out[0][2] = loop[2]; return 1; }
  for (int i = 0; i < n; ++i) {
    const int a = loop[(i + n - 1) % n], b = loop[i], c = loop[(i + 1) % n];
    if (mc_share_face(a, c)) continue;                   // the ear's new side (a, c) would lie in a cube face
    int rest[12], m = 0;
    for (int j = 0; j < n; ++j) if (j != i) rest[m++] = loop[j];
    const int sub = mc_triangulate(rest, m, out + 1);
    if (sub >= 0) { out[0][0] = a; out[0][1] = b; out[0][2] = c; return sub + 1; }
  }
  return -1;
}

static int build_mc_table(unsigned char (&ntri)[256], unsigned char (&tri)[256][MC_MAXT * 3]) {
  static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
  static const int edge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  static const int fcorner[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 2, 6, 7}, {0, 3, 7, 4}, {1, 2, 6, 5}};
  static const int fedge[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 9, 4, 8}, {2, 10, 6, 11}, {3, 11, 7, 8}, {1, 10, 5, 9}};
  int worst = 0;
  for (int cfg = 0; cfg < 256; ++cfg) {
    int nb[12][2], deg[12];
    for (int e = 0; e < 12; ++e) { deg[e] = 0; nb[e][0] = nb[e][1] = -1; }
    auto link = [&](int e0, int e1) { nb[e0][deg[e0]++] = e1; nb[e1][deg[e1]++] = e0; };
    for (int f = 0; f < 6; ++f) {
      int in[4], crossed[4], nc = 0;
      for (int k = 0; k < 4; ++k) in[k] = (cfg >> fcorner[f][k]) & 1;
      for (int k = 0; k < 4; ++k) { crossed[k] = in[k] != in[(k + 1) & 3]; nc += crossed[k]; }   // edge k joins corner k, k+1
      if (nc == 2) {
        int a = -1, b = -1;
        for (int k = 0; k < 4; ++k) if (crossed[k]) { if (a < 0) a = k; else b = k; }
        link(fedge[f][a], fedge[f][b]);
      } else if (nc == 4) {
        for (int k = 0; k < 4; ++k) if (in[k]) link(fedge[f][(k + 3) & 3], fedge[f][k]);          // cut off inside corner k
      }
    }
    int nt = 0;
    bool used[12] = {false};
    for (int e0 = 0; e0 < 12; ++e0) {
      if (deg[e0] == 0 || used[e0]) continue;
      if (deg[e0] != 2) return -1;
      int loop[12], n = 0, prev = -1, cur = e0;
      do {
        loop[n++] = cur; used[cur] = true;
        const int nxt = nb[cur][0] != prev ? nb[cur][0] : nb[cur][1];
        prev = cur; cur = nxt;
      } while (cur != e0 && n < 12);
      // orientation: Newell normal of the loop (vertices at the edge midpoints) against the in -> out directions of its edges
      double nrm[3] = {0, 0, 0}, dir[3] = {0, 0, 0};
      for (int i = 0; i < n; ++i) {
        double p[3], q[3];
        for (int d = 0; d < 3; ++d) {
          p[d] = 0.5 * (corner[edge[loop[i]][0]][d] + corner[edge[loop[i]][1]][d]);
          q[d] = 0.5 * (corner[edge[loop[(i + 1) % n]][0]][d] + corner[edge[loop[(i + 1) % n]][1]][d]);
        }
        nrm[0] += (p[1] - q[1]) * (p[2] + q[2]); nrm[1] += (p[2] - q[2]) * (p[0] + q[0]); nrm[2] += (p[0] - q[0]) * (p[1] + q[1]);
        const int c0 = edge[loop[i]][0], c1 = edge[loop[i]][1];
        const int cin = ((cfg >> c0) & 1) ? c0 : c1, cout = cin == c0 ? c1 : c0;
        for (int d = 0; d < 3; ++d) dir[d] += corner[cout][d] - corner[cin][d];
      }
      const bool flip = nrm[0] * dir[0] + nrm[1] * dir[1] + nrm[2] * dir[2] < 0.0;
      int tl[12][3];
      const int ntl = mc_triangulate(loop, n, tl);
      if (ntl < 0) return -1;
      for (int i = 0; i < ntl; ++i) {
        if (nt >= MC_MAXT) return -1;
        tri[cfg][3 * nt + 0] = (unsigned char)tl[i][0];
        tri[cfg][3 * nt + 1] = (unsigned char)tl[i][flip ? 2 : 1];
        tri[cfg][3 * nt + 2] = (unsigned char)tl[i][flip ? 1 : 2];
        ++nt;
      }
    }
    ntri[cfg] = (unsigned char)nt;
    worst = nt > worst ? nt : worst;
  }
  return worst;
}

// The table is built once per process (host side); the __constant__ copies are PER DEVICE, so the upload is tracked
// per device index (hipMemcpyToSymbol writes the current device only) under a mutex.
static int upload_mc_table() {
  static std::mutex mu;
  static int built = 0;                         // 0: not yet, 1: done, -1: failed
  static unsigned char ntri[256];
  static unsigned char tri[256][MC_MAXT * 3];
  static bool uploaded[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  if (built == 0) {
    for (auto& row : tri) for (auto& v : row) v = 0;
    built = build_mc_table(ntri, tri) < 0 ? -1 : 1;
  }
  if (built != 1) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (uploaded[dev]) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_mc_ntri), ntri, sizeof(ntri)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_mc_tri), tri, sizeof(tri)) != hipSuccess) return -1;
  uploaded[dev] = true;
  return 1;
}

template <bool MC>
static int extract(int B, const float* d_sdf, int n, float level, float cube_radius, int* d_offsets, int* d_tri_count,
                   float* d_tris, int max_tris, void* stream) {
  if (B <= 0 || n < 2 || d_sdf == nullptr || d_offsets == nullptr || d_tri_count == nullptr || d_tris == nullptr ||
      max_tris <= 0) { hm_set_error("hm_extract_surface: bad argument"); return -1; }
  if (MC && upload_mc_table() != 1) { hm_set_error("marching-cubes table generation failed"); return -2; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nc = n - 1, ncell = nc * nc * nc;
  dim3 grid((ncell + 255) / 256, B);
  hipLaunchKernelGGL((k_mt_count<MC>), grid, dim3(256), 0, st, d_sdf, n, level, d_offsets);
  hipLaunchKernelGGL(k_mt_scan, dim3(B), dim3(1024), 0, st, d_offsets, ncell, d_tri_count);
  hipLaunchKernelGGL((k_mt_emit<MC>), grid, dim3(256), 0, st, d_sdf, n, level, cube_radius, d_offsets, d_tris, max_tris);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

// d_sdf [B][n^3] (index (ix*n + iy)*n + iz), d_offsets scratch [B][(n-1)^3] ints, d_tri_count [B] out,
// d_tris [B][max_tris][9] out (three xyz vertices per triangle, object frame, scaled by cube_radius).
extern "C" int hm_extract_surface(int B, const float* d_sdf, int n, float level, float cube_radius,
                                  int* d_offsets, int* d_tri_count, float* d_tris, int max_tris, void* stream) {
  return extract<false>(B, d_sdf, n, level, cube_radius, d_offsets, d_tri_count, d_tris, max_tris, stream);
}

// the same with marching cubes (at most 8 triangles per cell)
extern "C" int hm_extract_surface_mc(int B, const float* d_sdf, int n, float level, float cube_radius,
                                     int* d_offsets, int* d_tri_count, float* d_tris, int max_tris, void* stream) {
  return extract<true>(B, d_sdf, n, level, cube_radius, d_offsets, d_tri_count, d_tris, max_tris, stream);
}
