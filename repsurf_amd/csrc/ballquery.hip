// ballquery.hip — radius neighbour lists for gfx950 (wave64).
//
// What it computes: query_ball_point(cuda=False) of the reference
// (classification/modules/pointnet2_utils.py:78-99):
//   d[s,k]  = square_distance(new_xyz[s], xyz[k])            (expanded formula, :15-25)
//   inside  = not (d > radius2)
//   idx[s,:] = first `nsample` inside k in ascending order, padded with the first one.
// The reference CUDA kernel (classification/modules/pointops/src/ballquery/ballquery_cuda_kernel.cu:47-80)
// runs one thread per centre and makes every thread re-read the whole cloud from global
// memory with direct-difference distances.  Here:
//   * a workgroup stages its cloud ONCE into LDS as SoA x/y/z/|p|^2 (coalesced global read,
//     conflict-free ds_read_b32 by consecutive lanes), blocks of one cloud are mapped onto one
//     XCD so the 12 KB cloud is fetched from HBM once per L2;
//   * one wave serves a centre: the 64 lanes test 64 consecutive points per step, `v_cmp`
//     writes the hit mask straight into an SGPR pair (the ballot), `v_mbcnt` turns it into each
//     hit's output slot (a wavefront prefix scan), `s_bcnt1` advances the count, and the loop
//     leaves as soon as nsample slots are filled -- so the ascending-index order of the
//     reference falls out of the lane order with no sort;
//   * QI centres are processed against each LDS chunk at once, so one set of 4 LDS reads
//     feeds QI distance evaluations (centre coordinates are wave-uniform -> SGPR operands).
// Algorithmic bytes per launch: 4*(3*b*n + 3*b*m + b*m*nsample); brute-force work is
// 6 VALU per (centre, point) pair, so at small b the kernel is launch/VALU-bound, not HBM-bound
// (DESIGN.md §5).
#include "rs_common.h"

namespace {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;
constexpr int BQ_QI = 4;            // centres evaluated together per wave
constexpr int BQ_LDS_POINTS = 4096; // 64 KB of LDS

template <bool USE_LDS>
__global__ void __launch_bounds__(BQ_THREADS)
ballquery_kernel(int b, int n, int m, float radius2, int nsample, int qpw, int blocks_per_cloud,
                 const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                 int *__restrict__ idx, int *__restrict__ cnt_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *sx = lds, *sy = lds + n, *sz = lds + 2 * n, *sp = lds + 3 * n;

  int cloud, chunk;
  rs_xcd_remap(blockIdx.x, b, blocks_per_cloud, cloud, chunk);
  const float *pts = xyz + (size_t)cloud * n * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  if (USE_LDS) {
    for (int p = tid; p < n; p += BQ_THREADS) {
      const float x = pts[p * 3 + 0], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
      sx[p] = x; sy[p] = y; sz[p] = z;
      sp[p] = rs_sqnorm(x, y, z);
    }
    __syncthreads();
  }

  const int q_begin = (chunk * BQ_WAVES + wave) * qpw;
  const int nchunks = (n + 63) >> 6;

  for (int q0 = q_begin; q0 < min(m, q_begin + qpw); q0 += BQ_QI) {
    float qx[BQ_QI], qy[BQ_QI], qz[BQ_QI], qq[BQ_QI];
    int cnt[BQ_QI], first[BQ_QI];
    bool live[BQ_QI];
#pragma unroll
    for (int i = 0; i < BQ_QI; ++i) {
      const int q = min(q0 + i, m - 1);
      const float *c = new_xyz + ((size_t)cloud * m + q) * 3;
      qx[i] = rs_uniform(c[0]); qy[i] = rs_uniform(c[1]); qz[i] = rs_uniform(c[2]);
      qq[i] = rs_sqnorm(qx[i], qy[i], qz[i]);
      cnt[i] = 0; first[i] = 0;
      live[i] = (q0 + i) < min(m, q_begin + qpw);
    }
    for (int c = 0; c < nchunks; ++c) {
      const int p = c * 64 + lane;
      float x, y, z, pp;
      if (USE_LDS) {
        const int pc = min(p, n - 1);
        x = sx[pc]; y = sy[pc]; z = sz[pc]; pp = sp[pc];
      } else {
        const int pc = min(p, n - 1);
        x = pts[pc * 3 + 0]; y = pts[pc * 3 + 1]; z = pts[pc * 3 + 2];
        pp = rs_sqnorm(x, y, z);
      }
      bool any_live = false;
#pragma unroll
      for (int i = 0; i < BQ_QI; ++i) {
        if (!live[i]) continue;                                   // wave-uniform
        const float d = rs_sqdist_expanded(qx[i], qy[i], qz[i], qq[i], x, y, z, pp);
        const bool inside = !(d > radius2) && (p < n);
        const unsigned long long mask = __ballot(inside);
        if (mask) {                                               // wave-uniform branch
          const int slot = cnt[i] + rs_mbcnt(mask);
          int *row = idx + ((size_t)cloud * m + (q0 + i)) * nsample;
          if (inside && slot < nsample) row[slot] = p;
          if (cnt[i] == 0) first[i] = c * 64 + (__ffsll((long long)mask) - 1);
          cnt[i] += __popcll(mask);
          if (cnt[i] >= nsample) live[i] = false;                 // row full: stop scanning for it
        }
        any_live |= live[i];
      }
      if (!any_live) break;
    }
    // pad with the first hit (pointnet2_utils.py:92-94); an empty ball yields zeros
#pragma unroll
    for (int i = 0; i < BQ_QI; ++i) {
      if ((q0 + i) >= min(m, q_begin + qpw)) continue;
      const int filled = min(cnt[i], nsample);
      int *row = idx + ((size_t)cloud * m + (q0 + i)) * nsample;
      for (int s = filled + lane; s < nsample; s += 64) row[s] = first[i];
      // distinct neighbours in the row (the rest is padding with the first one); an empty ball counts as one slot
      if (cnt_out && lane == 0) cnt_out[(size_t)cloud * m + (q0 + i)] = filled > 0 ? filled : 1;
    }
  }
}

}  // namespace

extern "C" int rs_ballquery(int b, int n, int m, float radius2, int nsample, const float *new_xyz,
                            const float *xyz, int *idx, int *cnt, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "rs_ballquery: negative size");
  if (b == 0 || m == 0 || nsample == 0) return RS_OK;
  RS_REQUIRE(n > 0, "rs_ballquery: empty cloud");
  RS_REQUIRE(new_xyz && xyz && idx, "rs_ballquery: null pointer");
  // centres per wave: keep >= ~2 workgroups per CU when the problem allows it
  int qpw = 8;
  while (qpw > BQ_QI && (long long)b * rs_cdiv(m, BQ_WAVES * qpw) < 512) qpw >>= 1;
  const int blocks_per_cloud = rs_cdiv(m, BQ_WAVES * qpw);
  const dim3 grid(b * blocks_per_cloud), block(BQ_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (n <= BQ_LDS_POINTS) {
    hipLaunchKernelGGL(ballquery_kernel<true>, grid, block, (size_t)n * 16, st, b, n, m, radius2, nsample,
                       qpw, blocks_per_cloud, new_xyz, xyz, idx, cnt);
  } else {
    hipLaunchKernelGGL(ballquery_kernel<false>, grid, block, 0, st, b, n, m, radius2, nsample, qpw,
                       blocks_per_cloud, new_xyz, xyz, idx, cnt);
  }
  RS_CHECK_LAUNCH("rs_ballquery");
  return RS_OK;
}
