// scene_knn.hip — k nearest neighbours inside ONE large cloud (whole-scene inference, N ~ 1e5 .. 1e6) for gfx950.
//
// What it computes: pointops.knnquery (segmentation/modules/pointops/functions/pointops.py:114-130 ->
// src/knnquery/knnquery_cuda_kernel.cu:65-108) for a single cloud, i.e. what the reference's whole-scene helpers call
// (segmentation/util/utils.py:235-245 `pc_median_filter_gpu`, segmentation/tool/test_s3dis.py:203-232): for every query
// the `nsample` rows with the smallest (direct-difference) squared distance, ascending by (distance, row).  The reference
// kernel is one thread per query scanning all N rows: 1e12 distance evaluations for a 1M-point room.  rs_knnquery_offset
// does the same scan 4 lanes per query through LDS tiles (0.9e12 pair tests per second: ~1 s at 1M); this file adds a
// uniform-grid search that evaluates only the rows of the query's 27 neighbouring cells:
//   rs_scene_cells    cell index of every row (cell = floor((x - lo) / edge), clamped) + a histogram (global atomics)
//   (rs_exclusive_scan of the histogram -> first sorted position of every cell)
//   rs_scene_scatter  rows scattered into cell-sorted order as float4 (x, y, z, row); order inside a cell is arbitrary,
//                     the result does not depend on it: candidates enter the list in (distance, row) order
//   rs_scene_knn      one thread per query: the 9 contiguous x-ranges of the 3 x 3 x 3 block, top-K list in registers,
//                     lexicographic insertion.  The list is PROVEN complete when the K-th distance is below the cell edge
//                     (a row outside the block is at least one cell edge away from a query inside the grid); otherwise the
//                     query is flagged and the caller runs the exact scan (rs_knnquery_offset) for it.  Same arithmetic
//                     (dx*dx + dy*dy) + dz*dz, same tie rule as the scan kernel: bit-identical lists and distances
//                     (tests/test_seg_gpu.py::test_scene_grid_knn_equals_scan).
#include "rs_common.h"
#include <math.h>

namespace {

constexpr int SK_THREADS = 256;

struct SceneGrid {
  float lo[3], hi[3];
  float inv;        // 1 / cell edge
  int g[3];
};

__device__ __forceinline__ int scene_cell(const SceneGrid &gr, float x, float y, float z, int &cx, int &cy, int &cz) {
  cx = min(gr.g[0] - 1, max(0, (int)((x - gr.lo[0]) * gr.inv)));
  cy = min(gr.g[1] - 1, max(0, (int)((y - gr.lo[1]) * gr.inv)));
  cz = min(gr.g[2] - 1, max(0, (int)((z - gr.lo[2]) * gr.inv)));
  return (cz * gr.g[1] + cy) * gr.g[0] + cx;
}

__global__ void __launch_bounds__(SK_THREADS)
scene_cells_kernel(int n, const float *__restrict__ xyz, SceneGrid gr, int *__restrict__ cell_of, int *__restrict__ counts) {
  const int i = blockIdx.x * SK_THREADS + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  const int c = scene_cell(gr, xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2], cx, cy, cz);
  cell_of[i] = c;
  atomicAdd(&counts[c], 1);
}

__global__ void __launch_bounds__(SK_THREADS)
scene_scatter_kernel(int n, const float *__restrict__ xyz, const int *__restrict__ cell_of, int *__restrict__ cursor,
                     float4 *__restrict__ sorted) {
  const int i = blockIdx.x * SK_THREADS + threadIdx.x;
  if (i >= n) return;
  const int pos = atomicAdd(&cursor[cell_of[i]], 1);
  sorted[pos] = make_float4(xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2], __int_as_float(i));
}

template <int K>
__device__ __forceinline__ void scene_insert(float (&bd)[K], int (&bi)[K], float d, int p) {
  if (d < bd[K - 1] || (d == bd[K - 1] && p < bi[K - 1])) {
    bd[K - 1] = d; bi[K - 1] = p;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool sw = bd[j] < bd[j - 1] || (bd[j] == bd[j - 1] && bi[j] < bi[j - 1]);
      const float td = bd[j]; const int ti = bi[j];
      bd[j] = sw ? bd[j - 1] : td; bi[j] = sw ? bi[j - 1] : ti;
      bd[j - 1] = sw ? td : bd[j - 1]; bi[j - 1] = sw ? ti : bi[j - 1];
    }
  }
}

template <int K>
__global__ void __launch_bounds__(SK_THREADS)
scene_knn_kernel(int m, int nsample, const float *__restrict__ q_xyz, SceneGrid gr, float accept_d2,
                 const int *__restrict__ starts, const float4 *__restrict__ sorted, int *__restrict__ idx,
                 float *__restrict__ dist2, int *__restrict__ flag) {
  const int q = blockIdx.x * SK_THREADS + threadIdx.x;
  if (q >= m) return;
  const float qx = q_xyz[(size_t)q * 3], qy = q_xyz[(size_t)q * 3 + 1], qz = q_xyz[(size_t)q * 3 + 2];
  float bd[K]; int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = 3.0e38f; bi[j] = 0x7fffffff; }
  int cx, cy, cz;
  scene_cell(gr, qx, qy, qz, cx, cy, cz);
  // a query outside the rows' bounding box sits outside its (clamped) cell: the one-cell-edge argument does not hold
  const bool inside = qx >= gr.lo[0] && qy >= gr.lo[1] && qz >= gr.lo[2] && qx <= gr.hi[0] && qy <= gr.hi[1] && qz <= gr.hi[2];
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gr.g[0] - 1);
  for (int z = max(cz - 1, 0); z <= min(cz + 1, gr.g[2] - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, gr.g[1] - 1); ++y) {
      const int cb = (z * gr.g[1] + y) * gr.g[0];
      const int j1 = starts[cb + x1 + 1];
      for (int j = starts[cb + x0]; j < j1; ++j) {
        const float4 c = sorted[j];
        const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
        const float d = (dx * dx + dy * dy) + dz * dz;       // knnquery_cuda_kernel.cu:94
        scene_insert<K>(bd, bi, d, __float_as_int(c.w));
      }
    }
  // complete iff nsample rows were found and the nsample-th is closer than any row outside the block can be
  const float kth = bd[min(nsample, K) - 1];
  const bool ok = inside && kth < accept_d2;
  flag[q] = ok ? 0 : 1;
  if (ok) {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j < nsample) { idx[(size_t)q * nsample + j] = bi[j]; dist2[(size_t)q * nsample + j] = bd[j]; }
  }
}

int scene_grid_of(const float *lo, const float *hi, float cell, const int *g, SceneGrid &gr, const char *who) {
  RS_REQUIRE(lo && hi && g && cell > 0.f && g[0] > 0 && g[1] > 0 && g[2] > 0, "%s: bad grid", who);
  RS_REQUIRE((long long)g[0] * g[1] * g[2] <= (1LL << 26), "%s: more than 2^26 cells", who);
  for (int a = 0; a < 3; ++a) { gr.lo[a] = lo[a]; gr.hi[a] = hi[a]; gr.g[a] = g[a]; }
  gr.inv = 1.0f / cell;
  return RS_OK;
}

}  // namespace

// lo, hi (3 HOST floats each): the rows' bounding box; cell: cell edge; g (3 HOST ints): cells per axis.
// counts (g0*g1*g2 + 1 ints) must be zero.
extern "C" int rs_scene_cells(int n, const float *xyz, const float *lo, const float *hi, float cell, const int *g, int *cell_of,
                              int *counts, void *stream) {
  RS_REQUIRE(n >= 0, "rs_scene_cells: negative size");
  if (n == 0) return RS_OK;
  RS_REQUIRE(xyz && cell_of && counts, "rs_scene_cells: null pointer");
  SceneGrid gr;
  int rc = scene_grid_of(lo, hi, cell, g, gr, "rs_scene_cells");
  if (rc != RS_OK) return rc;
  hipLaunchKernelGGL(scene_cells_kernel, dim3(rs_cdiv(n, SK_THREADS)), dim3(SK_THREADS), 0, (hipStream_t)stream, n, xyz, gr, cell_of, counts);
  RS_CHECK_LAUNCH("rs_scene_cells");
  return RS_OK;
}

// cursor: a copy of the cells' first positions (advanced in place); sorted: n float4 = (x, y, z, row as int bits)
extern "C" int rs_scene_scatter(int n, const float *xyz, const int *cell_of, int *cursor, float *sorted, void *stream) {
  RS_REQUIRE(n >= 0, "rs_scene_scatter: negative size");
  if (n == 0) return RS_OK;
  RS_REQUIRE(xyz && cell_of && cursor && sorted, "rs_scene_scatter: null pointer");
  hipLaunchKernelGGL(scene_scatter_kernel, dim3(rs_cdiv(n, SK_THREADS)), dim3(SK_THREADS), 0, (hipStream_t)stream, n, xyz, cell_of,
                     cursor, reinterpret_cast<float4 *>(sorted));
  RS_CHECK_LAUNCH("rs_scene_scatter");
  return RS_OK;
}

// idx / dist2 (m, nsample) are written for the queries whose flag comes back 0; flag 1 = run the exact scan for this query.
extern "C" int rs_scene_knn(int m, int nsample, const float *queries, const float *lo, const float *hi, float cell, const int *g,
                            const int *starts, const float *sorted, int *idx, float *dist2, int *flag, void *stream) {
  RS_REQUIRE(m >= 0 && nsample > 0, "rs_scene_knn: bad size");
  if (m == 0) return RS_OK;
  RS_REQUIRE(nsample <= 32, "rs_scene_knn: nsample=%d exceeds 32 (use rs_knnquery_offset)", nsample);
  RS_REQUIRE(queries && starts && sorted && idx && dist2 && flag, "rs_scene_knn: null pointer");
  SceneGrid gr;
  int rc = scene_grid_of(lo, hi, cell, g, gr, "rs_scene_knn");
  if (rc != RS_OK) return rc;
  const float edge = cell * (1.0f - 2e-3f);            // margin for the rounding of (x - lo) / cell at a cell face
  const float accept = edge * edge;
  const dim3 grid(rs_cdiv(m, SK_THREADS)), block(SK_THREADS);
  const float4 *s4 = reinterpret_cast<const float4 *>(sorted);
  hipStream_t st = (hipStream_t)stream;
  if (nsample <= 4) hipLaunchKernelGGL(scene_knn_kernel<4>, grid, block, 0, st, m, nsample, queries, gr, accept, starts, s4, idx, dist2, flag);
  else if (nsample <= 8) hipLaunchKernelGGL(scene_knn_kernel<8>, grid, block, 0, st, m, nsample, queries, gr, accept, starts, s4, idx, dist2, flag);
  else if (nsample <= 16) hipLaunchKernelGGL(scene_knn_kernel<16>, grid, block, 0, st, m, nsample, queries, gr, accept, starts, s4, idx, dist2, flag);
  else hipLaunchKernelGGL(scene_knn_kernel<32>, grid, block, 0, st, m, nsample, queries, gr, accept, starts, s4, idx, dist2, flag);
  RS_CHECK_LAUNCH("rs_scene_knn");
  return RS_OK;
}
