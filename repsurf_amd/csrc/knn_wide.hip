// Wide kNN (nsample > 64): the reference operators accept up to 200 neighbours per query
// (classification/modules/pointops/src/knnquery/knnquery_cuda_kernel.cu:21-22: best[200]) and 100 for the heap / packed
// variants (classification/.../knnquery_heap_cuda_kernel.cu:67-68, segmentation/.../knnquery_cuda_kernel.cu:86-87); no
// shipped model asks for more than 32, so this path buys contract width, not speed.  rs_knnquery / rs_knnquery_offset
// route nsample > 64 here; any nsample is accepted.
//
// One wave per query, selection by REPEATED MINIMUM: output slot j is the smallest (distance, row) pair that is
// lexicographically greater than slot j-1's.  A pass costs n/64 distance evaluations per lane + one wave reduction, nothing
// is stored per candidate, and the output is ascending by (distance, row) by construction -- the order of the register
// kernels (rs_knnquery: expanded-formula distances, ties by index; rs_knnquery_offset: direct differences, strict '<'
// replacement = ties in ascending row order, 1e10 / first-row padding when the cloud is smaller than nsample).
#include "rs_common.h"

namespace {

constexpr int KW_THREADS = 256;       // 4 queries per workgroup

// order-preserving map float -> uint (expanded-formula distances can be slightly negative)
__device__ __forceinline__ unsigned kw_key(float d) {
  const unsigned b = __float_as_uint(d);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

template <bool EXPANDED>
__device__ __forceinline__ void kw_select(const float *__restrict__ pts, int start, int end, float qx, float qy, float qz,
                                          int nsample, int *__restrict__ oidx, float *__restrict__ odist, int idx_base) {
  const int lane = rs_lane();
  const float qq = rs_sqnorm(qx, qy, qz);
  unsigned last_key = 0;
  int last_row = -1;
  bool first = true;
  int found = 0;
  for (int j = 0; j < nsample; ++j) {
    unsigned best_key = 0xFFFFFFFFu;
    int best_row = 0x7FFFFFFF;
    float best_d = 0.f;
    for (int p = start + lane; p < end; p += 64) {
      const float px = pts[(size_t)p * 3], py = pts[(size_t)p * 3 + 1], pz = pts[(size_t)p * 3 + 2];
      float d;
      if (EXPANDED) {
        d = rs_sqdist_expanded(qx, qy, qz, qq, px, py, pz, rs_sqnorm(px, py, pz));
      } else {
        const float dx = qx - px, dy = qy - py, dz = qz - pz;
        d = (dx * dx + dy * dy) + dz * dz;
      }
      const unsigned key = kw_key(d);
      const bool after = first || key > last_key || (key == last_key && p > last_row);
      // rows of a lane ascend, so '<' keeps the lowest row among equal keys
      if (after && key < best_key) { best_key = key; best_row = p; best_d = d; }
    }
    const unsigned wkey = rs_wave_min_u32(best_key);
    const unsigned wrow = rs_wave_min_u32(best_key == wkey ? (unsigned)best_row : 0x7FFFFFFFu);
    if (wrow == 0x7FFFFFFFu) break;                        // the cloud holds fewer than nsample rows
    const unsigned long long owner = __ballot(best_key == wkey && (unsigned)best_row == wrow);
    const float wd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_d), __ffsll((long long)owner) - 1));
    if (lane == 0) {
      oidx[j] = (int)wrow - idx_base;
      if (odist) odist[j] = wd;
    }
    last_key = wkey; last_row = (int)wrow; first = false;
    found = j + 1;
  }
  if (!EXPANDED)                                           // knnquery_cuda_kernel.cu:86-87: untouched slots stay (1e10, first row)
    for (int j = found + lane; j < nsample; j += 64) {
      oidx[j] = start - idx_base;
      if (odist) odist[j] = 1e10f;
    }
}

__global__ void __launch_bounds__(KW_THREADS)
knn_wide_dense_kernel(int b, int n, int m, int nsample, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                      int *__restrict__ idx, float *__restrict__ dist2) {
  const long long q = (long long)blockIdx.x * (KW_THREADS / 64) + (threadIdx.x >> 6);
  if (q >= (long long)b * m) return;
  const int cloud = (int)(q / m);
  const float *qp = new_xyz + q * 3;
  kw_select<true>(xyz, cloud * n, cloud * n + n, qp[0], qp[1], qp[2], nsample, idx + q * nsample,
                  dist2 ? dist2 + q * nsample : nullptr, cloud * n);
}

__global__ void __launch_bounds__(KW_THREADS)
knn_wide_packed_kernel(int m, int nsample, int b, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                       const int *__restrict__ offset, const int *__restrict__ new_offset, int *__restrict__ idx,
                       float *__restrict__ dist2) {
  const int q = blockIdx.x * (KW_THREADS / 64) + (threadIdx.x >> 6);
  if (q >= m) return;
  int c = 0;
  while (c < b - 1 && q >= new_offset[c]) ++c;             // get_bt_idx, knnquery_cuda_kernel.cu:51-62
  const int start = c ? offset[c - 1] : 0, end = offset[c];
  const float *qp = new_xyz + (size_t)q * 3;
  kw_select<false>(xyz, start, end, qp[0], qp[1], qp[2], nsample, idx + (size_t)q * nsample,
                   dist2 ? dist2 + (size_t)q * nsample : nullptr, 0);
}

}  // namespace

void rs_launch_knn_wide_dense(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx,
                              float *dist2, hipStream_t st) {
  const long long queries = (long long)b * m;
  hipLaunchKernelGGL(knn_wide_dense_kernel, dim3(rs_cdiv(queries, KW_THREADS / 64)), dim3(KW_THREADS), 0, st, b, n, m,
                     nsample, xyz, new_xyz, idx, dist2);
}

void rs_launch_knn_wide_packed(int m, int nsample, int b, const float *xyz, const float *new_xyz, const int *offset,
                               const int *new_offset, int *idx, float *dist2, hipStream_t st) {
  hipLaunchKernelGGL(knn_wide_packed_kernel, dim3(rs_cdiv(m, KW_THREADS / 64)), dim3(KW_THREADS), 0, st, m, nsample, b,
                     xyz, new_xyz, offset, new_offset, idx, dist2);
}
