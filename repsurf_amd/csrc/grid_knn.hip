// grid_knn.hip — k nearest neighbours of packed batches through per-cloud uniform grids, for gfx950 (round 4).
//
// What it computes: pointops.knnquery (segmentation/modules/pointops/functions/pointops.py:114-130 ->
// src/knnquery/knnquery_cuda_kernel.cu:65-108) -- for every query the `nsample` rows of ITS cloud with the smallest
// direct-difference squared distance (dx*dx + dy*dy) + dz*dz (:93), ascending, an equal distance never displacing an earlier
// row (strict '<', :96), lists shorter than nsample padded with (1e10, first row of the cloud) (:86-87): the same lists, bit
// for bit, as rs_knnquery_offset (csrc/seg_geom.hip), which scans the whole cloud per query.
//
// Why: the segmentation step's geometry stage spends 1.2 ms per step in those scans (16 x 4096 points: 65 536 queries x 4096
// candidates for the umbrella fans, 16 384 x 4096 for the first grouping, ...) on a second stream BESIDE the network, and the
// network runs 0.74 ms slower for it (4.19 ms with the geometry stage, 3.45 ms without: profiles/r04/geometry_contention.txt).
// A query's neighbours sit in a ball that holds a fraction of a percent of its cloud.
//
//   rs_knn_grid_build   one workgroup per cloud: bounding box -> grid (cubic cells of edge e, ~`per_cell` rows per cell, at most
//                       RS_KNN_GRID_CELLS cells) -> histogram (LDS atomics) -> exclusive scan -> rows scattered into cell-sorted
//                       order as float4 (x, y, z, row).  The order inside a cell is arbitrary and the result does not depend on
//                       it: candidates are ranked by (distance, row).
//   rs_knn_grid_query   one thread per query: the cells at Chebyshev distance 0, 1, 2, ... from the query's cell, ring by ring
//                       (every ring is (2r+1)^2 rows of cells, contiguous x-ranges of the sorted array).  After ring r every row
//                       NOT visited is at least r cell edges away along some axis, so the list is complete as soon as its
//                       nsample-th distance is below (r e)^2 -- or the rings have covered the whole grid, which is the scan.
//                       A workgroup takes queries of one cloud and stages that cloud's sorted rows in LDS (<= 4096 rows).
//                       (First version: a max-heap per thread in LDS for nsample = 32 -- every sift-down is a chain of five LDS
//                       round trips that the whole wave waits for whenever one lane inserts: 401 us against the scan's 453 us.)
// No fallback launch and no host read-back: every query ends by one of the two conditions above.
#include "rs_common.h"
#include "umbrella_fan.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int GK_CELLS = RS_KNN_GRID_CELLS;
constexpr int GK_BUILD_T = 1024;
constexpr int GK_LDS_ROWS = 4096;                // rows of one cloud staged in LDS (64 KB)
constexpr float GK_MARGIN = 1.0f - 2.0e-3f;      // rounding of (x - lo) / e at a cell face (<= 2.5e-4 e at 1024 cells per axis)

struct CloudGrid {          // 16 words per cloud in the `grid` workspace
  float lo[3]; float inv;   // cell of x: min(g - 1, max(0, (int)((x - lo) * inv)))
  float edge;               // e * GK_MARGIN
  int g[3];
  int first, rows;          // the cloud's rows [first, first + rows)
  float slack;              // how far the EXPANDED distance formula of the classification path (rs_sqdist_expanded) can be from
                            // the true squared distance inside this cloud's bounding box: 64 * 2^-24 * (largest norm)^2 (bound: 22)
  int pad[5];
};
static_assert(sizeof(CloudGrid) == 64, "CloudGrid is 16 words");

__device__ __forceinline__ unsigned ordered(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float unordered(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ int cell_coord(float x, float lo, float inv, int g) {
  const float t = fminf(fmaxf((x - lo) * inv, 0.f), (float)(g - 1));     // (saturates instead of converting a huge value)
  return (int)t;
}

__global__ void __launch_bounds__(GK_BUILD_T)
grid_build_kernel(const float *__restrict__ xyz, const int *__restrict__ offset, float per_cell, float4 *__restrict__ sorted,
                  int *__restrict__ starts, CloudGrid *__restrict__ grid) {
  __shared__ int hist[GK_CELLS];
  __shared__ unsigned red[6][GK_BUILD_T / 64];
  __shared__ int wsum[GK_BUILD_T / 64];
  __shared__ CloudGrid G;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int first = c ? offset[c - 1] : 0, n = offset[c] - first;
  const float *src = xyz + (size_t)first * 3;
  // ---- bounding box
  unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int p = tid; p < n; p += GK_BUILD_T)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const unsigned o = ordered(src[(size_t)p * 3 + a]); mn[a] = min(mn[a], o); mx[a] = max(mx[a], o); }
#pragma unroll
  for (int a = 0; a < 3; ++a) { mn[a] = rs_wave_min_u32(mn[a]); mx[a] = rs_wave_max_u32(mx[a]); }
  if (lane == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; }
  __syncthreads();
  if (tid == 0) {
    float lo[3], ext[3], big = 0.f;
    for (int a = 0; a < 3; ++a) {
      unsigned l = 0xffffffffu, h = 0u;
      for (int w = 0; w < GK_BUILD_T / 64; ++w) { l = min(l, red[a][w]); h = max(h, red[3 + a][w]); }
      lo[a] = n > 0 ? unordered(l) : 0.f;
      ext[a] = n > 0 ? unordered(h) - lo[a] : 0.f;
      big = fmaxf(big, ext[a]);
    }
    if (!(big > 0.f)) big = 1.f;                               // one point / coincident points: any grid will do
    float vol = 1.f;
    for (int a = 0; a < 3; ++a) vol *= fmaxf(ext[a], big * 1e-3f);    // flat clouds: the thin axis gets one layer of cells
    float want = fminf(fmaxf((float)n / fmaxf(per_cell, 0.25f), 1.f), (float)GK_CELLS);
    float e = cbrtf(vol / want);
    int g[3];
    for (int it = 0; it < 64; ++it) {
      long long cells = 1;
      for (int a = 0; a < 3; ++a) { g[a] = (int)fminf(fmaxf(ceilf(ext[a] / e), 1.f), 1024.f); cells *= g[a]; }
      if (cells <= GK_CELLS && ext[0] / e <= 1024.f && ext[1] / e <= 1024.f && ext[2] / e <= 1024.f) break;
      e *= 1.1f;
    }
    for (int a = 0; a < 3; ++a) { G.lo[a] = lo[a]; G.g[a] = g[a]; }
    G.inv = 1.0f / e;
    G.edge = e * GK_MARGIN;
    G.first = first; G.rows = n;
    float r2 = 0.f;
    for (int a = 0; a < 3; ++a) { const float m = fmaxf(fabsf(lo[a]), fabsf(lo[a] + ext[a])); r2 += m * m; }
    G.slack = 64.f * 5.9604645e-8f * r2;
    for (int k = 0; k < 5; ++k) G.pad[k] = 0;
    grid[c] = G;
  }
  for (int i = tid; i < GK_CELLS; i += GK_BUILD_T) hist[i] = 0;
  __syncthreads();
  const int ncell = G.g[0] * G.g[1] * G.g[2];
  auto cell_of = [&](int p) {
    const float x = src[(size_t)p * 3], y = src[(size_t)p * 3 + 1], z = src[(size_t)p * 3 + 2];
    return (cell_coord(z, G.lo[2], G.inv, G.g[2]) * G.g[1] + cell_coord(y, G.lo[1], G.inv, G.g[1])) * G.g[0] + cell_coord(x, G.lo[0], G.inv, G.g[0]);
  };
  for (int p = tid; p < n; p += GK_BUILD_T) atomicAdd(&hist[cell_of(p)], 1);
  __syncthreads();
  // ---- exclusive scan of the histogram: 4 consecutive cells per thread, wave scan, wave totals through LDS
  constexpr int PER = GK_CELLS / GK_BUILD_T;
  int v[PER], run = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { v[k] = run; run += hist[tid * PER + k]; }
  int inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - run;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int *st = starts + (size_t)c * (GK_CELLS + 1);
  __syncthreads();                                   // every thread has read its counts: the histogram becomes the cursors
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    hist[i] = base + v[k];
    if (i <= ncell) st[i] = first + base + v[k];     // (i == ncell: the end of the last cell = first + n)
  }
  if (ncell == GK_CELLS && tid == 0) st[GK_CELLS] = first + n;
  __syncthreads();
  for (int p = tid; p < n; p += GK_BUILD_T) {
    const int pos = atomicAdd(&hist[cell_of(p)], 1);
    sorted[(size_t)first + pos] = make_float4(src[(size_t)p * 3], src[(size_t)p * 3 + 1], src[(size_t)p * 3 + 2], __int_as_float(first + p));
  }
}

// (distance, row) as ONE unsigned 64-bit key: the order-preserving image of the distance's bits (ordered(): a float compares like
// the unsigned value, negative zero and the slightly negative results of the expanded distance formula included) in the high word,
// the row in the low one; "an equal distance never displaces an earlier row" is then a plain integer '<'.  (Written as d < e ||
// (d == e && p < q) every step of the insertion below was two branches over the exec mask: a 32-entry list took 2.5 us per
// insertion.)
typedef unsigned long long key_t;
__device__ __forceinline__ key_t make_key(float d, int row) { return ((key_t)ordered(d) << 32) | (key_t)(unsigned)row; }
__device__ __forceinline__ float key_dist(key_t k) { return unordered((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_row(key_t k) { return (int)(unsigned)(k & 0xffffffffULL); }
constexpr key_t KEY_SENTINEL = 0ULL;                               // below every real key
constexpr key_t KEY_INF = ~0ULL;                                   // above every real key

template <int K>
__device__ __forceinline__ void list_insert(key_t (&ls)[K], key_t k) {
  if (k < ls[K - 1]) {
    ls[K - 1] = k;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const key_t a = ls[j - 1], b = ls[j];
      const bool sw = b < a;
      ls[j - 1] = sw ? b : a;
      ls[j] = sw ? a : b;
    }
  }
}

// One workgroup = T consecutive queries of ONE cloud (blockIdx.y); the cloud's cell-sorted rows and cell starts are staged in LDS
// when they fit (`lds_rows` float4 / GK_CELLS + 1 ints), else read from global memory.  The list (capacity K >= nsample, ascending
// keys) lives in registers.  A candidate that beats the list's last entry is only PARKED (P slots per thread in LDS, column =
// thread: the bank is the lane); the wave inserts parked candidates together, when some lane's slots run out or a ring ends --
// candidate by candidate the K-step insertion ran whenever ANY of the 64 lanes had something to insert, i.e. for nearly every
// candidate.  All loops are wave-uniform (ring, row of cells, trip of four candidates); a lane whose list is complete walks
// empty ranges.  For lists of up to 16 entries (umbrella fans: 9, interpolation: 3), where there are many queries and the
// insertion is short.
// (Measured for 32-entry lists and not kept: this kernel, 345 us per launch whatever the number of queries -- ~115 insertions per
// lane, ~200 wave steps of a 31-deep compare-exchange chain, one wave per SIMD; and a two-pass form that first COUNTS the
// candidates into buckets of d^2 / edge^2 to bound the nsample-th distance and then inserts only ~45 of them: 668-1035 us, the
// walk itself is what costs -- ~180 VALU instructions per trip of four candidates issued by one wave per SIMD.  Lists longer
// than 16 entries go to grid_wave_kernel below.)
//
// UMB (classification): the fused umbrella-surface constructor of csrc/knn_umbrella.hip over the grid -- the queries are the cloud's
// own rows, the distance is the EXPANDED formula of query_knn_point (rs_sqdist_expanded: what the scan kernel evaluates, so the
// keys and with them the lists are the same, bit for bit), whose value can differ from the true squared distance by `slack`: a
// list is complete when its last distance is below (r e)^2 - slack.  The epilogue is umbrella_kernel's: the K - 1 neighbours
// after the nearest -> fan features (umbrella_fan.h), one 40 (K - 1)-byte row per point; idx = the lists (optional), dist2 unused.
template <int K, int T, int P, bool UMB>
__global__ void __launch_bounds__(T)
grid_query_kernel(int nsample, const float *__restrict__ new_xyz, const int *__restrict__ new_offset,
                  const float4 *__restrict__ sorted, const int *__restrict__ starts, const CloudGrid *__restrict__ grid,
                  int lds_rows, int *__restrict__ idx, float *__restrict__ dist2, const float *__restrict__ inv_sign,
                  float *__restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  key_t *park = reinterpret_cast<key_t *>(smem);                                          // [P][T]
  float4 *lrows = reinterpret_cast<float4 *>(smem + (size_t)P * T * 8);                    // [lds_rows]
  int *lst = reinterpret_cast<int *>(smem + (size_t)P * T * 8 + (size_t)lds_rows * 16);    // [GK_CELLS + 1] (only when lds_rows > 0)
  const int c = blockIdx.y, tid = threadIdx.x;
  const int qs = c ? new_offset[c - 1] : 0, qe = new_offset[c];
  const int q0 = qs + blockIdx.x * T;
  if (q0 >= qe) return;                                              // (workgroup-uniform)
  const CloudGrid G = grid[c];
  const int *st = starts + (size_t)c * (GK_CELLS + 1);
  const int gx = G.g[0], gy = G.g[1], gz = G.g[2];
  const bool staged = lds_rows > 0 && G.rows <= lds_rows;
  if (staged) {
    for (int p = tid; p < G.rows; p += T) lrows[p] = sorted[(size_t)G.first + p];
    for (int i = tid; i <= gx * gy * gz; i += T) lst[i] = st[i] - G.first;        // LDS positions
    __syncthreads();
  }
  const int q = q0 + tid;
  const bool mine = q < qe;
  const int qc = mine ? q : qe - 1;
  const float qx = new_xyz[(size_t)qc * 3], qy = new_xyz[(size_t)qc * 3 + 1], qz = new_xyz[(size_t)qc * 3 + 2];
  const int cx = cell_coord(qx, G.lo[0], G.inv, gx), cy = cell_coord(qy, G.lo[1], G.inv, gy), cz = cell_coord(qz, G.lo[2], G.inv, gz);
  // capacity K >= nsample: the first K - nsample entries are sentinels below every real key, so the list proper ends at entry
  // K - 1 whatever nsample is (a run-time index into the list is a round trip through scratch memory)
  key_t ls[K];
  const int lead = K - nsample;
  const key_t empty = UMB ? make_key(INFINITY, 0x7fffffff) : make_key(1e10f, G.first);   // csrc/knn_umbrella.hip knn_scan4 / knnquery_cuda_kernel.cu:86-87
  const float qq = rs_sqnorm(qx, qy, qz);
#pragma unroll
  for (int j = 0; j < K; ++j) ls[j] = j < lead ? KEY_SENTINEL : empty;
  int np = 0;

  auto flush = [&]() {
    for (int s = 0; __any(s < np); ++s)
      if (s < np) list_insert<K>(ls, park[s * T + tid]);
    np = 0;
  };
  // rows [j, j1) of the cloud's sorted array, four per trip behind one wait (clamped loads, results masked)
  auto walk = [&](int j, int j1, auto staged_) {
    constexpr bool LDS = decltype(staged_)::value;
    while (__any(j < j1)) {
      float4 v[4];
      const int last = max(j1 - 1, j);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = min(j + u, last);
        if constexpr (LDS) v[u] = lrows[jj]; else v[u] = sorted[jj];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float d;
        if constexpr (UMB) {
          d = rs_sqdist_expanded(qx, qy, qz, qq, v[u].x, v[u].y, v[u].z, rs_sqnorm(v[u].x, v[u].y, v[u].z));
        } else {
          const float dx = qx - v[u].x, dy = qy - v[u].y, dz = qz - v[u].z;
          d = (dx * dx + dy * dy) + dz * dz;                         // knnquery_cuda_kernel.cu:93 (no contraction: -ffp-contract=off)
        }
        const key_t k = make_key(d, __float_as_int(v[u].w));
        if (j + u < j1 && k < ls[K - 1]) { park[np * T + tid] = k; ++np; }
      }
      j += 4;
      if (__any(np > P - 4)) flush();
    }
  };
  // the (at most two) ranges of row (dz, dy) of ring r: a face row of the ring is one x-range, an inner row its two end cells
  auto row_ranges = [&](int r, int dz, int dy, bool active, int (&rg)[4], auto staged_) {
    constexpr bool LDS = decltype(staged_)::value;
    const int *cs = LDS ? lst : st;
    const int z = cz + dz, y = cy + dy;
    const bool in = active && z >= 0 && z < gz && y >= 0 && y < gy;
    const int cb = in ? (z * gy + y) * gx : 0;
    rg[0] = rg[1] = rg[2] = rg[3] = 0;
    if (dz == -r || dz == r || dy == -r || dy == r) {                 // (wave-uniform branch)
      if (in) { rg[0] = cs[cb + max(cx - r, 0)]; rg[1] = cs[cb + min(cx + r, gx - 1) + 1]; }
    } else {
      if (in && cx - r >= 0) { rg[0] = cs[cb + cx - r]; rg[1] = cs[cb + cx - r + 1]; }
      if (in && cx + r < gx) { rg[2] = cs[cb + cx + r]; rg[3] = cs[cb + cx + r + 1]; }
    }
  };
  // ring by ring, every candidate through the list, until the list is provably complete
  auto search = [&](auto staged_) {
    bool done = !mine;
    for (int r = 0; __any(!done); ++r) {
      // rows of the ring that can lie inside the grid for SOME query (every query's cell is inside it): the walk is bounded by the
      // grid, not by r, for a query far away from every row of its cloud
      const int z0 = max(-r, -(gz - 1)), z1 = min(r, gz - 1), y0 = max(-r, -(gy - 1)), y1 = min(r, gy - 1);
      int cur[4], nxt[4];
      row_ranges(r, z0, y0, !done, cur, staged_);
      for (int dz = z0; dz <= z1; ++dz)
        for (int dy = y0; dy <= y1; ++dy) {
          const bool more = dy < y1 || dz < z1;
          nxt[0] = nxt[1] = nxt[2] = nxt[3] = 0;
          if (more) row_ranges(r, dy < y1 ? dz : dz + 1, dy < y1 ? dy + 1 : y0, !done, nxt, staged_);   // the next row's first positions travel under this row's walk
          walk(cur[0], cur[1], staged_);
          walk(cur[2], cur[3], staged_);
#pragma unroll
          for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        }
      flush();
      if (!done) {
        const bool covered = cx - r <= 0 && cx + r >= gx - 1 && cy - r <= 0 && cy + r >= gy - 1 && cz - r <= 0 && cz + r >= gz - 1;
        const float reach = (float)r * G.edge;
        if (covered || key_dist(ls[K - 1]) < reach * reach - (UMB ? G.slack : 0.f)) done = true;   // every cell visited / every unvisited row is farther than the nsample-th
      }
    }
  };
  if (staged) search(std::true_type{}); else search(std::false_type{});
  if (!mine) return;
  if constexpr (UMB) {                                                // (nsample == K here)
    constexpr int G1 = K - 1;
    if (idx) {
#pragma unroll
      for (int l = 0; l < K; ++l) idx[(size_t)q * K + l] = key_row(ls[l]) - G.first;     // positions inside the cloud, like rs_umbrella_features
    }
    // offsets of the k-1 neighbours that follow the nearest (repsurface_utils.py:119-121)
    float ox[G1], oy[G1], oz[G1];
#pragma unroll
    for (int j = 0; j < G1; ++j) {
      const float *p = new_xyz + (size_t)key_row(ls[j + 1]) * 3;
      ox[j] = p[0] - qx; oy[j] = p[1] - qy; oz[j] = p[2] - qz;
    }
    rs_fan_features<G1, false, false>(ox, oy, oz, inv_sign ? inv_sign[c] : 1.f, feat + (size_t)q * (G1 * 10));
  } else {
    int *oi = idx + (size_t)q * nsample;
    float *od = dist2 ? dist2 + (size_t)q * nsample : nullptr;
#pragma unroll
    for (int l = 0; l < K; ++l)
      if (l >= lead) { oi[l - lead] = key_row(ls[l]); if (od) od[l - lead] = key_dist(ls[l]); }
  }
}

// ---- lists of 17 .. 64 entries: one WAVE per query ------------------------------------------------------------------------------
// There are few such queries (the centres of a grouping stage: 16 384 at the first level, a quarter of that at each next one) and
// each wants a long list: a thread per query leaves the chip idle behind a handful of waves that each walk a latency chain.
// Here the 64 lanes of a wave share ONE query: the ranges of a ring's rows of cells are wave-uniform, lane l evaluates candidate
// j + l of a range (consecutive LDS addresses), and the list is spread over the lanes -- lane i holds the i-th smallest key, lanes
// >= nsample hold +infinity.  Candidates that beat the current nsample-th key are compacted (ballot + prefix count) into a pending
// buffer of the wave; 64 pending keys are sorted across the lanes by a bitonic network (21 compare-exchange stages through
// lane shuffles), laid against the list in opposite order -- the elementwise minimum of an ascending and a descending sequence
// is a bitonic sequence holding the 64 smallest of the 128 -- and merged by 6 more stages.  A workgroup of 16 waves stages its
// cloud's rows once and each wave takes `qpw` queries; the result row leaves as one coalesced store per query.
constexpr int GW_T = 1024;               // 16 waves share one staged cloud: four per SIMD hide each other's shuffle / LDS latency
                                         // (4 waves per workgroup, one workgroup per CU by its LDS: 185-285 us for 16 384 queries)

__device__ __forceinline__ key_t shfl_xor_key(key_t k, int m) {
  const int lo = __shfl_xor((int)(unsigned)(k & 0xffffffffULL), m, 64), hi = __shfl_xor((int)(unsigned)(k >> 32), m, 64);
  return ((key_t)(unsigned)hi << 32) | (key_t)(unsigned)lo;
}
// ascending sort of one key per lane
__device__ __forceinline__ key_t wave_sort(key_t k, int lane) {
#pragma unroll
  for (int sz = 2; sz <= 64; sz <<= 1)
#pragma unroll
    for (int j = sz >> 1; j > 0; j >>= 1) {
      const key_t o = shfl_xor_key(k, j);
      const bool up = (lane & sz) == 0, lower = (lane & j) == 0;      // (sz = 64: every lane ascending)
      const bool take_min = up == lower;
      k = take_min ? (o < k ? o : k) : (o > k ? o : k);
    }
  return k;
}
// a bitonic sequence (one key per lane) -> ascending
__device__ __forceinline__ key_t wave_merge(key_t k, int lane) {
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const key_t o = shfl_xor_key(k, j);
    k = (lane & j) == 0 ? (o < k ? o : k) : (o > k ? o : k);
  }
  return k;
}

__global__ void __launch_bounds__(GW_T)
grid_wave_kernel(int nsample, int qpw, const float *__restrict__ new_xyz, const int *__restrict__ new_offset,
                 const float4 *__restrict__ sorted, const int *__restrict__ starts, const CloudGrid *__restrict__ grid,
                 int lds_rows, int *__restrict__ idx, float *__restrict__ dist2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = GW_T / 64;
  key_t *pend_all = reinterpret_cast<key_t *>(smem);                                       // [NW][128]
  float4 *lrows = reinterpret_cast<float4 *>(smem + (size_t)NW * 128 * 8);                 // [lds_rows]
  int *lst = reinterpret_cast<int *>(smem + (size_t)NW * 128 * 8 + (size_t)lds_rows * 16); // [GK_CELLS + 1]
  const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qs = c ? new_offset[c - 1] : 0, qe = new_offset[c];
  const int q0 = qs + blockIdx.x * NW * qpw;
  if (q0 >= qe) return;                                              // (workgroup-uniform)
  const CloudGrid G = grid[c];
  const int *st = starts + (size_t)c * (GK_CELLS + 1);
  const int gx = G.g[0], gy = G.g[1], gz = G.g[2];
  const bool staged = lds_rows > 0 && G.rows <= lds_rows;
  if (staged) {
    for (int p = tid; p < G.rows; p += GW_T) lrows[p] = sorted[(size_t)G.first + p];
    for (int i = tid; i <= gx * gy * gz; i += GW_T) lst[i] = st[i] - G.first;
    __syncthreads();
  }
  key_t *pend = pend_all + wave * 128;
  const key_t empty = make_key(1e10f, G.first);                      // knnquery_cuda_kernel.cu:86-87

  for (int qi = 0; qi < qpw; ++qi) {
    const int q = q0 + wave * qpw + qi;                              // (wave-uniform)
    if (q >= qe) break;
    const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
    const int cx = cell_coord(qx, G.lo[0], G.inv, gx), cy = cell_coord(qy, G.lo[1], G.inv, gy), cz = cell_coord(qz, G.lo[2], G.inv, gz);
    key_t mine = lane < nsample ? empty : KEY_INF;                   // lane i: the i-th smallest key so far
    key_t kth = empty;                                               // the nsample-th (wave-uniform)
    int npend = 0;                                                   // (wave-uniform)

    auto absorb = [&](int cnt) {                                     // the first `cnt` <= 64 pending keys into the list
      key_t b = lane < cnt ? pend[lane] : KEY_INF;
      b = wave_sort(b, lane);
      const key_t rev = shfl_xor_key(b, 63);                         // descending
      key_t m = rev < mine ? rev : mine;                             // bitonic: the 64 smallest of list + batch
      m = wave_merge(m, lane);
      mine = lane < nsample ? m : KEY_INF;
      const int lo = __builtin_amdgcn_readlane((int)(unsigned)(mine & 0xffffffffULL), nsample - 1);
      const int hi = __builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), nsample - 1);
      kth = ((key_t)(unsigned)hi << 32) | (key_t)(unsigned)lo;
    };
    auto range = [&](int j, int j1, auto staged_) {                  // rows [j, j1): wave-uniform bounds, 64 candidates per trip
      constexpr bool LDS = decltype(staged_)::value;
      for (int base = j; base < j1; base += 64) {
        const int jj = base + lane;
        const bool valid = jj < j1;
        float4 v;
        if constexpr (LDS) v = lrows[valid ? jj : j]; else v = sorted[valid ? jj : j];
        const float dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
        const float d = (dx * dx + dy * dy) + dz * dz;               // knnquery_cuda_kernel.cu:93 (no contraction: -ffp-contract=off)
        const key_t k = make_key(d, __float_as_int(v.w));
        const bool pass = valid && k < kth;
        const unsigned long long mask = __ballot(pass);
        if (mask) {
          if (pass) pend[npend + rs_mbcnt(mask)] = k;
          npend += __popcll(mask);
          if (npend >= 64) {                                          // a full batch: absorb it, keep the rest
            absorb(64);
            const key_t rest = (lane + 64 < npend) ? pend[lane + 64] : KEY_INF;
            pend[lane] = rest;
            npend -= 64;
          }
        }
      }
    };
    auto search = [&](auto staged_) {
      constexpr bool LDS = decltype(staged_)::value;
      const int *cs = LDS ? lst : st;
      for (int r = 0;; ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, gz - 1), y0 = max(cy - r, 0), y1 = min(cy + r, gy - 1);
        for (int z = z0; z <= z1; ++z)
          for (int y = y0; y <= y1; ++y) {
            const int cb = (z * gy + y) * gx;
            if (z == cz - r || z == cz + r || y == cy - r || y == cy + r) {
              range(cs[cb + max(cx - r, 0)], cs[cb + min(cx + r, gx - 1) + 1], staged_);
            } else {
              if (cx - r >= 0) range(cs[cb + cx - r], cs[cb + cx - r + 1], staged_);
              if (cx + r < gx) range(cs[cb + cx + r], cs[cb + cx + r + 1], staged_);
            }
          }
        if (npend > 0) { absorb(npend); npend = 0; }
        const bool covered = cx - r <= 0 && cx + r >= gx - 1 && cy - r <= 0 && cy + r >= gy - 1 && cz - r <= 0 && cz + r >= gz - 1;
        const float reach = (float)r * G.edge;
        if (covered || key_dist(kth) < reach * reach) break;         // every cell visited / every unvisited row is farther than the nsample-th
      }
    };
    if (staged) search(std::true_type{}); else search(std::false_type{});
    if (lane < nsample) {
      idx[(size_t)q * nsample + lane] = key_row(mine);
      if (dist2) dist2[(size_t)q * nsample + lane] = key_dist(mine);
    }
  }
}

}  // namespace

extern "C" int rs_knn_grid_build(int b, const float *xyz, const int *offset, float per_cell, float *sorted, int *starts,
                                 float *grid, void *stream) {
  RS_REQUIRE(b >= 0, "rs_knn_grid_build: negative size");
  if (b == 0) return RS_OK;
  RS_REQUIRE(xyz && offset && sorted && starts && grid, "rs_knn_grid_build: null pointer");
  RS_REQUIRE(per_cell > 0.f, "rs_knn_grid_build: per_cell must be positive");
  RS_REQUIRE(((uintptr_t)sorted % 16) == 0 && ((uintptr_t)grid % 16) == 0, "rs_knn_grid_build: sorted / grid must be 16-byte aligned");
  hipLaunchKernelGGL(grid_build_kernel, dim3(b), dim3(GK_BUILD_T), 0, (hipStream_t)stream, xyz, offset, per_cell,
                     reinterpret_cast<float4 *>(sorted), starts, reinterpret_cast<CloudGrid *>(grid));
  RS_CHECK_LAUNCH("rs_knn_grid_build");
  return RS_OK;
}

extern "C" int rs_knn_grid_query(int m, int nsample, int b, int max_queries, int max_rows, const float *new_xyz, const int *new_offset,
                                 const float *sorted, const int *starts, const float *grid, int *idx, float *dist2, void *stream) {
  RS_REQUIRE(m >= 0 && nsample >= 0 && b >= 0, "rs_knn_grid_query: negative size");
  if (m == 0 || nsample == 0 || b == 0) return RS_OK;
  RS_REQUIRE(nsample <= 64, "rs_knn_grid_query: nsample=%d exceeds 64 (rs_knnquery_offset takes those)", nsample);
  RS_REQUIRE(new_xyz && new_offset && sorted && starts && grid && idx, "rs_knn_grid_query: null pointer");
  RS_REQUIRE(max_queries >= 0 && max_rows >= 0, "rs_knn_grid_query: negative bound");
  if (max_queries <= 0 || max_queries > m) max_queries = m;           // unknown: one cloud may hold every query
  hipStream_t st = (hipStream_t)stream;
  const float4 *s4 = reinterpret_cast<const float4 *>(sorted);
  const CloudGrid *cg = reinterpret_cast<const CloudGrid *>(grid);
  // LDS: the rows of the largest cloud, at most GK_LDS_ROWS of them, + its cell starts (larger clouds are read in place)
  int lds_rows = max_rows > 0 && max_rows < GK_LDS_ROWS ? max_rows : GK_LDS_ROWS;
  // (a device with less LDS than the staged cloud + lists need: no staging, the rows are read in place -- ADVICE r4)
  if ((size_t)8 * 256 * 8 + (size_t)lds_rows * 16 + (size_t)(GK_CELLS + 1) * 4 > (size_t)rs_lds_limit()) lds_rows = 0;
  if (nsample > 16) {      // (short lists through this kernel: 65 536 queries, 9 entries 190-290 us against 122; 3 entries 157-174 against 74)
    // one wave per query, `qpw` queries per wave: enough waves to fill the chip (16 per CU) before a wave takes a second query
    int qpw = (int)(((long long)max_queries * b + 4095) / 4096);
    qpw = qpw < 1 ? 1 : (qpw > 64 ? 64 : qpw);
    const dim3 g(rs_cdiv(max_queries, (GW_T / 64) * qpw), b), t(GW_T);
    const size_t lds = (size_t)(GW_T / 64) * 128 * 8 + (size_t)lds_rows * 16 + (size_t)(GK_CELLS + 1) * 4;
    hipLaunchKernelGGL(grid_wave_kernel, g, t, lds, st, nsample, qpw, new_xyz, new_offset, s4, starts, cg, lds_rows, idx, dist2);
    RS_CHECK_LAUNCH("rs_knn_grid_query");
    return RS_OK;
  }
#define RS_GQ(K_, T_, P_) do {                                                                                              \
    const dim3 g(rs_cdiv(max_queries, T_), b), t(T_);                                                                         \
    const size_t lds = (size_t)P_ * T_ * 8 + (size_t)lds_rows * 16 + (size_t)(GK_CELLS + 1) * 4;                              \
    hipLaunchKernelGGL((grid_query_kernel<K_, T_, P_, false>), g, t, lds, st, nsample, new_xyz, new_offset, s4, starts, cg, lds_rows, idx, dist2, \
                       (const float *)nullptr, (float *)nullptr);                                                           \
  } while (0)
  if (nsample <= 3) RS_GQ(3, 256, 8);
  else if (nsample <= 9) RS_GQ(9, 256, 8);
  else RS_GQ(16, 256, 8);
#undef RS_GQ
  RS_CHECK_LAUNCH("rs_knn_grid_query");
  return RS_OK;
}

/* rs_umbrella_features through the grid: rs_knn_grid_build over the B clouds of n rows (offset = n, 2n, ...), then the fused
 * search + fan kernel.  Same outputs, bit for bit (the lists are the scan's). */
extern "C" int rs_umbrella_features_grid(int b, int n, int k, const float *xyz, const int *offset, const float *inv_sign, int *knn_idx,
                                         float *feat, float *sorted, int *starts, float *grid, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0, "rs_umbrella_features_grid: negative size");
  if (b == 0 || n == 0) return RS_OK;
  RS_REQUIRE(k == 5 || k == 9 || k == 13 || k == 17,
             "rs_umbrella_features_grid: k=%d not built (group_size+1 must be 5, 9, 13 or 17)", k);
  RS_REQUIRE(n >= k, "rs_umbrella_features_grid: cloud of %d points cannot supply %d neighbours", n, k);
  RS_REQUIRE(xyz && offset && feat && sorted && starts && grid, "rs_umbrella_features_grid: null pointer");
  RS_REQUIRE((long long)b * n < 2147483647LL, "rs_umbrella_features_grid: more than 2^31 rows");
  int rc = rs_knn_grid_build(b, xyz, offset, 1.5f, sorted, starts, grid, stream);
  if (rc != RS_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const float4 *s4 = reinterpret_cast<const float4 *>(sorted);
  const CloudGrid *cg = reinterpret_cast<const CloudGrid *>(grid);
  int lds_rows = n < GK_LDS_ROWS ? n : GK_LDS_ROWS;
  if ((size_t)8 * 256 * 8 + (size_t)lds_rows * 16 + (size_t)(GK_CELLS + 1) * 4 > (size_t)rs_lds_limit()) lds_rows = 0;      // (read the rows in place)
#define RS_GU(K_) do {                                                                                                       \
    constexpr int T_ = 256, P_ = 8;                                                                                           \
    const dim3 g(rs_cdiv(n, T_), b), t(T_);                                                                                   \
    const size_t lds = (size_t)P_ * T_ * 8 + (size_t)lds_rows * 16 + (size_t)(GK_CELLS + 1) * 4;                              \
    hipLaunchKernelGGL((grid_query_kernel<K_, T_, P_, true>), g, t, lds, st, K_, xyz, offset, s4, starts, cg, lds_rows, knn_idx, \
                       (float *)nullptr, inv_sign, feat);                                                                     \
  } while (0)
  switch (k) {
    case 5: RS_GU(5); break;
    case 9: RS_GU(9); break;
    case 13: RS_GU(13); break;
    default: RS_GU(17); break;
  }
#undef RS_GU
  RS_CHECK_LAUNCH("rs_umbrella_features_grid");
  return RS_OK;
}
