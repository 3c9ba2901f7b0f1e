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
#include <stdlib.h>
#include <math.h>

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

// ---- grid-accelerated variant ---------------------------------------------------------------------------------
// The scan above tests every (centre, point) pair: ~8 VALU operations per pair make it VALU-bound at a few per cent
// of the HBM roofline its 2.7 MB of traffic is priced against, whatever the batch.  Here a workgroup bins its cloud
// into a uniform grid in LDS (cell edge >= 1.001 r, at most BG_MAXG cells per axis; counting sort by cell) and a centre
// only tests the points of its 27 neighbouring cells -- with the SAME distance arithmetic and threshold, so the hit
// set is the brute-force one (a point inside the computed radius is at most r + 3e-6 away: three orders of magnitude
// inside the 1e-3 r margin of the cell edge, which also covers the rounding of the cell index) -- then puts the hits
// in ascending index order, which is the reference's "first nsample in index order"
// (classification/modules/pointnet2_utils.py:89-94).  Rows that overflow nsample select the nsample lowest indices
// by repeated minimum (dense clusters only).
constexpr int BG_THREADS = 256;
constexpr int BG_MAXG = 12;                              // cells per axis
constexpr int BG_MAXCELLS = BG_MAXG * BG_MAXG * BG_MAXG;
constexpr int BG_MAXN = 4096;                            // LDS: 24 n + 8 (BG_MAXCELLS + 1) bytes

__global__ void __launch_bounds__(BG_THREADS)
ballquery_grid_kernel(int b, int n, int m, float radius2, int nsample, int wpc, const float *__restrict__ new_xyz,
                      const float *__restrict__ xyz, int *__restrict__ idx, int *__restrict__ cnt_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *sx = lds, *sy = lds + n, *sz = lds + 2 * n, *sp = lds + 3 * n;
  int *order = reinterpret_cast<int *>(lds + 4 * n);          // point indices grouped by cell
  int *cellof = order + n;
  int *cstart = cellof + n;                                     // running starts, ncell + 1 entries
  int *cursor = cstart + BG_MAXCELLS + 1;
  __shared__ float red[6][BG_THREADS / 64];
  __shared__ int scan_tmp[BG_THREADS];

  int cloud, chunk;
  rs_xcd_remap(blockIdx.x, b, wpc, cloud, chunk);
  const float *pts = xyz + (size_t)cloud * n * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // A. stage the cloud, bounding box
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int p = tid; p < n; p += BG_THREADS) {
    const float x = pts[p * 3 + 0], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
    sx[p] = x; sy[p] = y; sz[p] = z; sp[p] = rs_sqnorm(x, y, z);
    lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
    hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = lo[a], h = hi[a];
    for (int off = 32; off > 0; off >>= 1) { l = fminf(l, __shfl_xor(l, off, 64)); h = fmaxf(h, __shfl_xor(h, off, 64)); }
    if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
  }
  __syncthreads();
  float ext = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = red[a][0], h = red[3 + a][0];
#pragma unroll
    for (int w = 1; w < BG_THREADS / 64; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
    lo[a] = l; hi[a] = h;
    ext = fmaxf(ext, h - l);
  }
  // B. grid geometry (identical in every thread): the centres are points of the cloud's bounding box too in the
  //    model, but any centre works -- its cell coordinate is clamped, and a clamped centre farther than one cell from
  //    the box cannot have neighbours inside it
  const float cell = fmaxf(sqrtf(radius2) * 1.001f, ext * (1.0001f / BG_MAXG));
  const float inv = 1.0f / cell;
  int g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = min(BG_MAXG, (int)((hi[a] - lo[a]) * inv) + 1);
  const int ncell = g[0] * g[1] * g[2];

  // C. histogram
  for (int c = tid; c < ncell; c += BG_THREADS) cstart[c] = 0;
  __syncthreads();
  for (int p = tid; p < n; p += BG_THREADS) {
    const int cx = min(g[0] - 1, (int)((sx[p] - lo[0]) * inv));
    const int cy = min(g[1] - 1, (int)((sy[p] - lo[1]) * inv));
    const int cz = min(g[2] - 1, (int)((sz[p] - lo[2]) * inv));
    const int c = (cz * g[1] + cy) * g[0] + cx;
    cellof[p] = c;
    atomicAdd(&cstart[c], 1);
  }
  __syncthreads();
  // D. exclusive scan of the counts: each thread owns a run of consecutive cells
  const int per = (ncell + BG_THREADS - 1) / BG_THREADS;
  int run = 0;
  for (int k = 0; k < per; ++k) { const int c = tid * per + k; if (c < ncell) run += cstart[c]; }
  scan_tmp[tid] = run;
  __syncthreads();
  for (int off = 1; off < BG_THREADS; off <<= 1) {
    const int v = tid >= off ? scan_tmp[tid - off] : 0;
    __syncthreads();
    scan_tmp[tid] += v;
    __syncthreads();
  }
  int base = scan_tmp[tid] - run;
  for (int k = 0; k < per; ++k) {
    const int c = tid * per + k;
    if (c < ncell) { const int v = cstart[c]; cstart[c] = base; cursor[c] = base; base += v; }
  }
  if (tid == 0) cstart[ncell] = n;
  __syncthreads();
  // E. counting sort (order inside a cell is arbitrary: hits are sorted by index afterwards)
  for (int p = tid; p < n; p += BG_THREADS) order[atomicAdd(&cursor[cellof[p]], 1)] = p;
  __syncthreads();

  // F. centres of this workgroup, 256 at a time: every thread fills ITS row in LDS (stride nsample + 1: conflict-free),
  //    then the workgroup writes the 256 finished rows -- contiguous in the output -- with coalesced stores.  (Rows
  //    written slot by slot from 64 lanes touch 64 cache lines per instruction: 4x slower end to end.)
  int *rowbuf = cursor + BG_MAXCELLS;
  const int rstride = nsample + 1;
  int *row = rowbuf + tid * rstride;
  const int qper = (m + wpc - 1) / wpc;
  const int q_beg = chunk * qper, q_end = min(m, (chunk + 1) * qper);
  for (int qb = q_beg; qb < q_end; qb += BG_THREADS) {
    const int q = qb + tid;
    int count = 0;
    if (q < q_end) {
      const float *c = new_xyz + ((size_t)cloud * m + q) * 3;
      const float qx = c[0], qy = c[1], qz = c[2], qq = rs_sqnorm(qx, qy, qz);
      // cell of the centre (floor, clamped in float so that far-away centres stay in int range)
      const float fx = fminf(fmaxf((qx - lo[0]) * inv, -2.f), (float)BG_MAXG + 2.f);
      const float fy = fminf(fmaxf((qy - lo[1]) * inv, -2.f), (float)BG_MAXG + 2.f);
      const float fz = fminf(fmaxf((qz - lo[2]) * inv, -2.f), (float)BG_MAXG + 2.f);
      const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g[0] - 1);
      const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g[1] - 1);
      const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g[2] - 1);
      if (x0 <= x1) {
        for (int z = z0; z <= z1; ++z)
          for (int y = y0; y <= y1; ++y) {
            const int cb = (z * g[1] + y) * g[0];
            for (int j = cstart[cb + x0]; j < cstart[cb + x1 + 1]; ++j) {      // the 3 x-neighbours are contiguous
              const int p = order[j];
              const float d = rs_sqdist_expanded(qx, qy, qz, qq, sx[p], sy[p], sz[p], sp[p]);
              if (!(d > radius2)) {
                if (count < nsample) row[count] = p;
                ++count;
              }
            }
          }
      }
      if (count > nsample) {
        // overflow: the nsample lowest indices among the hits, by repeated minimum over the same candidates
        int last = -1;
        for (int s = 0; s < nsample; ++s) {
          int best = 0x7fffffff;
          for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
              const int cb = (z * g[1] + y) * g[0];
              for (int j = cstart[cb + x0]; j < cstart[cb + x1 + 1]; ++j) {
                const int p = order[j];
                if (p > last && p < best) {
                  const float d = rs_sqdist_expanded(qx, qy, qz, qq, sx[p], sy[p], sz[p], sp[p]);
                  if (!(d > radius2)) best = p;
                }
              }
            }
          row[s] = best;
          last = best;
        }
      } else {
        // insertion sort of the `count` hits by index (4-15 on average at the model's radii)
        for (int i = 1; i < count; ++i) {
          const int v = row[i];
          int j = i - 1;
          while (j >= 0 && row[j] > v) { row[j + 1] = row[j]; --j; }
          row[j + 1] = v;
        }
        const int first = count > 0 ? row[0] : 0;            // pad with the lowest hit; an empty ball yields zeros
        for (int s = count; s < nsample; ++s) row[s] = first;
      }
      if (cnt_out) cnt_out[(size_t)cloud * m + q] = count > 0 ? min(count, nsample) : 1;
    }
    __syncthreads();
    const int nq = min(BG_THREADS, q_end - qb);
    int *dst = idx + ((size_t)cloud * m + qb) * nsample;
    for (int e = tid; e < nq * nsample; e += BG_THREADS) {
      const int ql = e / nsample, sl = e - ql * nsample;
      dst[e] = rowbuf[ql * rstride + sl];
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int rs_ballquery(int b, int n, int m, float radius2, int nsample, const float *new_xyz,
                            const float *xyz, int *idx, int *cnt, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "rs_ballquery: negative size");
  if (b == 0 || m == 0 || nsample == 0) return RS_OK;
  RS_REQUIRE(n > 0, "rs_ballquery: empty cloud");
  RS_REQUIRE(new_xyz && xyz && idx, "rs_ballquery: null pointer");
  hipStream_t st = (hipStream_t)stream;
  // -1 (default): the grid variant where it wins -- many clouds per launch, >= 1024 points, short rows (measured:
  // 212 us against 610 us at B=2048 x 1024 x 512, nsample 32; at B=32 the scan is faster, 25 us against 28 us, and
  // for 512-point clouds with r=0.4 the 27 cells hold a fifth of the cloud); 0 / 1 force the scan / the grid
  static const int grid_mode = getenv("RS_BALLQUERY_GRID") ? atoi(getenv("RS_BALLQUERY_GRID")) : -1;
  const bool grid_ok = n <= BG_MAXN && n >= 64 && nsample <= 64;
  const bool use_grid = grid_mode > 0 || (grid_mode < 0 && (long long)b * m >= 65536 && n >= 1024 && nsample <= 32);
  if (use_grid && grid_ok) {
    // workgroups per cloud: one when the batch alone fills the chip, more (each rebuilds the small grid) otherwise
    int wpc = 1;
    while (wpc < 8 && (long long)b * wpc < 512 && m / (wpc * 2) >= 32) wpc *= 2;
    const size_t lds = (size_t)n * 24 + sizeof(int) * (2 * BG_MAXCELLS + 1 + (size_t)BG_THREADS * (nsample + 1));
    hipLaunchKernelGGL(ballquery_grid_kernel, dim3(b * wpc), dim3(BG_THREADS), lds, st, b, n, m, radius2, nsample, wpc,
                       new_xyz, xyz, idx, cnt);
    RS_CHECK_LAUNCH("rs_ballquery");
    return RS_OK;
  }
  // centres per wave: keep >= ~2 workgroups per CU when the problem allows it
  int qpw = 8;
  while (qpw > BQ_QI && (long long)b * rs_cdiv(m, BQ_WAVES * qpw) < 512) qpw >>= 1;
  const int blocks_per_cloud = rs_cdiv(m, BQ_WAVES * qpw);
  const dim3 grid(b * blocks_per_cloud), block(BQ_THREADS);
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
