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
#include <type_traits>

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


// ---- cell list, second design (round 2) --------------------------------------------------------------------------
// ballquery_grid_kernel above spends its time in latency, not work: a thread walks 9 rows of cells one after the other,
// every candidate costs a dependent LDS chain (order[j] -> x/y/z/|p|^2), a wave pays the LONGEST row of its 64 lanes nine
// times over, hits are insertion-sorted in LDS (a dependent read-compare-write chain as long as the busiest lane's
// count squared) and 72 KB of LDS per 256 threads leave 2 waves per SIMD to hide any of it.  Here:
//   * the cloud is counting-sorted BY CELL into an array of float4 (x, y, z, |p|^2) + an index array, so a candidate is
//     one 16-byte LDS read at address j and the three x-neighbour cells of a row are one contiguous range;
//   * a thread takes its centre's <= 9 rows FOUR candidates per step behind one wait, the hit bookkeeping runs only when
//     a ballot says some lane has a hit (the flattened one-candidate-per-step stream of the first version of this
//     kernel is kept for the rare overflow rows);
//   * points stay in registers between the bounding-box, histogram and scatter passes (no second read), the prefix sum of
//     the cell counts is a wave scan + one LDS hop (2 barriers instead of 16);
//   * a thread keeps up to BC_HCAP hits in its LDS row and sorts them in REGISTERS with a Batcher network (19 / 63
//     compare-exchanges for <= 8 / <= 16 hits: straight-line v_min/v_max, no LDS round trips);
//   * rows with more hits (dense clusters) are redone by their WAVE, one row at a time (round 3): 64 lanes over the row's
//     candidates, ranks by counting; rows with more than 64 hits by a cooperative repeated minimum;
//   * the other rows leave through one coalesced sweep over the (centres x nsample) block, padding expanded on the fly.
// Same distance arithmetic and threshold as the scan kernel: bit-identical neighbour lists (tests/test_geometry_gpu.py).
// Phase costs at B = 2048 x 1024 points x 512 centres, r = 0.2 (RS_BALLQUERY_DBG builds, profiles/r02/): see DESIGN.md §5.
constexpr int BC_HCAP = 16;
constexpr int BC_SLACK = 4;                              // float4 entries behind the sorted cloud a 4-candidate step may read

__device__ __forceinline__ void bc_cx(int &a, int &b) { const int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
template <int N> __device__ __forceinline__ void bc_sort(int (&v)[16]) {
  constexpr unsigned char P8[19][2] = {{0,1},{2,3},{0,2},{1,3},{1,2},{4,5},{6,7},{4,6},{5,7},{5,6},{0,4},{2,6},{2,4},{1,5},{3,7},{3,5},{1,2},{3,4},{5,6}};
  constexpr unsigned char P16[63][2] = {{0,1},{2,3},{0,2},{1,3},{1,2},{4,5},{6,7},{4,6},{5,7},{5,6},{0,4},{2,6},{2,4},{1,5},{3,7},{3,5},{1,2},{3,4},{5,6},
    {8,9},{10,11},{8,10},{9,11},{9,10},{12,13},{14,15},{12,14},{13,15},{13,14},{8,12},{10,14},{10,12},{9,13},{11,15},{11,13},{9,10},{11,12},{13,14},
    {0,8},{4,12},{4,8},{2,10},{6,14},{6,10},{2,4},{6,8},{10,12},{1,9},{5,13},{5,9},{3,11},{7,15},{7,11},{3,5},{7,9},{11,13},{1,2},{3,4},{5,6},{7,8},{9,10},{11,12},{13,14}};
  if constexpr (N == 8) {
#pragma unroll
    for (int i = 0; i < 19; ++i) bc_cx(v[P8[i][0]], v[P8[i][1]]);
  } else {
#pragma unroll
    for (int i = 0; i < 63; ++i) bc_cx(v[P16[i][0]], v[P16[i][1]]);
  }
}

template <int BC_THREADS, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(BC_THREADS, WAVES_PER_SIMD)
ballquery_cells_kernel(int b, int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
                       const float *__restrict__ xyz, int *__restrict__ idx, int *__restrict__ cnt_out, int dbg) {
  constexpr int BC_PP = BG_MAXN / BC_THREADS;                  // points a thread carries in registers (8 / 16)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4 *sp4 = reinterpret_cast<float4 *>(lds);               // (x, y, z, |p|^2) sorted by cell (+ BC_SLACK entries)
  int *sid = reinterpret_cast<int *>(lds + 4 * (n + BC_SLACK));
  int *cstart = sid + n;                                        // ncell + 1 running starts
  int *hcount = cstart + BG_MAXCELLS + 1;                       // one hit count per thread / centre
  int *arena = hcount + BC_THREADS;                             // BC_THREADS / 64 rows of 64: a wave's cooperative overflow row
  int *cursor = arena + BC_THREADS;                             // scatter cursors (build) ...
  unsigned short *hits = reinterpret_cast<unsigned short *>(cursor);   // ... then BC_THREADS rows of BC_HCAP + 1 hits (16-bit: n <= 4096)
  __shared__ float red[6][BC_THREADS / 64];
  __shared__ int wsum[BC_THREADS / 64];

  const int cloud = blockIdx.x;
  const float *pts = xyz + (size_t)cloud * n * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // A. this thread's points (registers), bounding box
  float px[BC_PP], py[BC_PP], pz[BC_PP];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int k = 0; k < BC_PP; ++k) {
    const int p = tid + k * BC_THREADS;
    if (p < n) {
      px[k] = pts[p * 3 + 0]; py[k] = pts[p * 3 + 1]; pz[k] = pts[p * 3 + 2];
      lo[0] = fminf(lo[0], px[k]); lo[1] = fminf(lo[1], py[k]); lo[2] = fminf(lo[2], pz[k]);
      hi[0] = fmaxf(hi[0], px[k]); hi[1] = fmaxf(hi[1], py[k]); hi[2] = fmaxf(hi[2], pz[k]);
    } else { px[k] = py[k] = pz[k] = 0.f; }
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
    for (int w = 1; w < BC_THREADS / 64; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
    lo[a] = l; hi[a] = h;
    ext = fmaxf(ext, h - l);
  }
  // B. grid geometry: the same formulas as ballquery_grid_kernel (cell edge >= 1.001 r)
  const float cell = fmaxf(sqrtf(radius2) * 1.001f, ext * (1.0001f / BG_MAXG));
  const float inv = 1.0f / cell;
  int g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = min(BG_MAXG, (int)((hi[a] - lo[a]) * inv) + 1);
  const int ncell = g[0] * g[1] * g[2];
  auto cell_of = [&](float x, float y, float z) {
    const int cx = min(g[0] - 1, (int)((x - lo[0]) * inv));
    const int cy = min(g[1] - 1, (int)((y - lo[1]) * inv));
    const int cz = min(g[2] - 1, (int)((z - lo[2]) * inv));
    return (cz * g[1] + cy) * g[0] + cx;
  };

  // C. histogram
  for (int c = tid; c < ncell; c += BC_THREADS) cstart[c] = 0;
  __syncthreads();
  int pc[BC_PP];
#pragma unroll
  for (int k = 0; k < BC_PP; ++k)
    if (tid + k * BC_THREADS < n) { pc[k] = cell_of(px[k], py[k], pz[k]); atomicAdd(&cstart[pc[k]], 1); }
  __syncthreads();
  // D. exclusive scan of the counts: a run of consecutive cells per thread, wave scan, one LDS hop
  const int per = (ncell + BC_THREADS - 1) / BC_THREADS;
  int run = 0;
  for (int k = 0; k < per; ++k) { const int c = tid * per + k; if (c < ncell) run += cstart[c]; }
  int incl = run;
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - run;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int k = 0; k < per; ++k) {
    const int c = tid * per + k;
    if (c < ncell) { const int v = cstart[c]; cstart[c] = base; cursor[c] = base; base += v; }
  }
  if (tid == 0) cstart[ncell] = n;
  __syncthreads();
  // E. scatter the points into their cells (order inside a cell is arbitrary: hits are sorted by index afterwards)
#pragma unroll
  for (int k = 0; k < BC_PP; ++k) {
    const int p = tid + k * BC_THREADS;
    if (p < n) {
      const int pos = atomicAdd(&cursor[pc[k]], 1);
      sp4[pos] = make_float4(px[k], py[k], pz[k], rs_sqnorm(px[k], py[k], pz[k]));
      sid[pos] = p;
    }
  }
  __syncthreads();
  if (dbg == 1) return;                                         // (experiment: cost of the build alone)

  // F. centres, BC_THREADS at a time (the cursors are spent: their LDS now holds the hit rows)
  unsigned short *row = hits + tid * (BC_HCAP + 1);
  for (int qb = 0; qb < m; qb += BC_THREADS) {
    const int q = qb + tid;
    int count = 0;
    float qx = 0.f, qy = 0.f, qz = 0.f, qq = 0.f;
    int x0 = 1, x1 = 0, y0 = 0, z0 = 0, ny = 0, nrows = 0;
    if (q < m) {
      const float *c = new_xyz + ((size_t)cloud * m + q) * 3;
      qx = c[0]; qy = c[1]; qz = c[2]; qq = rs_sqnorm(qx, qy, qz);
      const float fx = fminf(fmaxf((qx - lo[0]) * inv, -2.f), (float)BG_MAXG + 2.f);
      const float fy = fminf(fmaxf((qy - lo[1]) * inv, -2.f), (float)BG_MAXG + 2.f);
      const float fz = fminf(fmaxf((qz - lo[2]) * inv, -2.f), (float)BG_MAXG + 2.f);
      const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
      x0 = max(cx - 1, 0); x1 = min(cx + 1, g[0] - 1);
      y0 = max(cy - 1, 0); const int y1 = min(cy + 1, g[1] - 1);
      z0 = max(cz - 1, 0); const int z1 = min(cz + 1, g[2] - 1);
      ny = y1 - y0 + 1;
      nrows = (x0 <= x1 && ny > 0 && z1 >= z0) ? ny * (z1 - z0 + 1) : 0;
    }
    // Main pass: the centre's <= 9 rows of cells, FOUR candidates per step.  The row ranges are fetched up front (18
    // independent LDS reads), a step issues four 16-byte candidate reads behind one wait and evaluates them branch-free;
    // the (rare: ~5 hits among ~28 candidates) hit bookkeeping runs only when some lane of the wave has a hit (one
    // ballot).  A wave pays max-over-lanes ceil(len / 4) steps per row (~2.5) instead of one step per candidate of its
    // busiest lane (~43), at ~45 instead of ~40 instructions per step.
    {
      int jb[9], je[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int rz = r / 3, ry = r - rz * 3;                   // (ry, rz) enumerate a 3 x 3 block; clipped rows are empty
        const bool on = ry < ny && rz * ny + ry < nrows && ry + y0 < g[1] && rz + z0 < g[2];
        const int cb = ((z0 + (on ? rz : 0)) * g[1] + (y0 + (on ? ry : 0))) * g[0];
        jb[r] = on ? cstart[cb + x0] : 0;
        je[r] = on ? cstart[cb + x1 + 1] : 0;
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        int j = jb[r];
        const int jend = je[r];
        while (__ballot(j < jend)) {
          float4 c4[4];
          bool hit[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) c4[i] = sp4[j + i];                 // j <= jend <= n (finished lanes stand still below): at most BC_SLACK entries past the cloud, never valid
          bool any = false;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d = rs_sqdist_expanded(qx, qy, qz, qq, c4[i].x, c4[i].y, c4[i].z, c4[i].w);
            hit[i] = (j + i < jend) && !(d > radius2);
            any = any || hit[i];
          }
          if (__ballot(any)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (hit[i]) {
                if (count < BC_HCAP) row[count] = (unsigned short)sid[j + i];
                ++count;
              }
          }
          j = j < jend ? j + 4 : j;      // a lane whose row is exhausted stays inside it while the wave finishes the longest row
        }
      }
      // (tried, round 3: reading a slot only where it holds a candidate -- exec-masked ds_read_b128, 27 of ~80 slots -- and
      //  storing a hit's POSITION with the sid lookup deferred to the sort phase: walk 69.5 -> 70.8 us, sort 6 -> 9.6 us,
      //  launch 93 -> 101 us; the LDS pipe is not relieved by masked lanes and the deferred lookup is a second dependent
      //  round trip in front of the sorting network)
    }
    if (dbg == 2) { if (count == 12345) idx[0] = count; continue; }      // (experiment: build + walk)
    if (q < m && cnt_out) cnt_out[(size_t)cloud * m + q] = count > 0 ? min(count, nsample) : 1;
    // sort the hits by index in registers (wave-uniform choice of the network: 8 or 16 wide)
    {
      const int kept = min(count, BC_HCAP);
      const int wmax = __builtin_amdgcn_readfirstlane((int)rs_wave_max_u32((unsigned)kept));
      if (wmax > 1) {
        int v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (i < 8 || wmax > 8) ? ((i < kept) ? row[i] : 0x7fffffff) : 0x7fffffff;
        if (wmax > 8) bc_sort<16>(v); else bc_sort<8>(v);
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if ((i < 8 || wmax > 8) && i < kept) row[i] = (unsigned short)v[i];
      }
    }
    hcount[tid] = q < m ? count : 0;
    __syncthreads();
    if (dbg == 3) continue;                                       // (experiment: build + walk + sort)
    // coalesced write-out of the rows with <= BC_HCAP hits; padding = the lowest hit (an empty ball yields zeros)
    const int nq = min(BC_THREADS, m - qb);
    int *dst = idx + ((size_t)cloud * m + qb) * nsample;
    if (dbg != 5 && (nsample & 3) == 0) {
      // 16 bytes per thread and trip: a row is nsample / 4 groups of four slots; a wave stores 1 KB of consecutive output
      const int gpr = nsample >> 2;
      int4 *dst4 = reinterpret_cast<int4 *>(dst);
      for (int e = tid; e < nq * gpr; e += BC_THREADS) {
        const int ql = e / gpr, g4 = (e - ql * gpr) << 2;
        const int c = hcount[ql];
        if (c <= BC_HCAP) {
          const unsigned short *h = hits + ql * (BC_HCAP + 1);
          const int take = min(c, nsample);
          const int first = c > 0 ? h[0] : 0;
          int4 v;
          v.x = g4 + 0 < take ? h[min(g4 + 0, BC_HCAP)] : first;
          v.y = g4 + 1 < take ? h[min(g4 + 1, BC_HCAP)] : first;
          v.z = g4 + 2 < take ? h[min(g4 + 2, BC_HCAP)] : first;
          v.w = g4 + 3 < take ? h[min(g4 + 3, BC_HCAP)] : first;
          dst4[e] = v;
        }
      }
    } else if (dbg != 5) {
      int ql = tid / nsample, sl = tid - ql * nsample;
      const int dq = BC_THREADS / nsample, ds = BC_THREADS - dq * nsample;
      for (int e = tid; e < nq * nsample; e += BC_THREADS) {
        const int c = hcount[ql];
        if (c <= BC_HCAP) {
          const unsigned short *h = hits + ql * (BC_HCAP + 1);
          const int take = min(c, nsample);
          dst[e] = c > 0 ? h[sl < take ? sl : 0] : 0;
        }
        ql += dq; sl += ds;
        if (sl >= nsample) { sl -= nsample; ++ql; }
      }
    }
    // Rows with more than BC_HCAP hits (dense clusters; ~1e-5 of the rows of a uniform cloud): the WAVE redoes them
    // together, one row at a time -- 64 lanes over the row's candidates, hits through a ballot into the wave's arena, ranks by
    // counting (<= 64 hits) or the nsample lowest indices by repeated minimum (more).  A single thread walking and
    // insertion-sorting such a row in LDS took ~12 us of dependent latency, and the launch ended with the unluckiest
    // workgroup: 13 us of a 113 us launch for a dozen rows in a million (profiles/r02/ballquery_cells_phase_costs.txt).
    if (dbg != 4) {
      unsigned long long todo = __ballot(q < m && count > BC_HCAP);
      int *mine = arena + wave * 64;
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const float cqx = __shfl(qx, src, 64), cqy = __shfl(qy, src, 64), cqz = __shfl(qz, src, 64), cqq = __shfl(qq, src, 64);
        const int cx0 = __shfl(x0, src, 64), cx1 = __shfl(x1, src, 64), cy0 = __shfl(y0, src, 64), cz0 = __shfl(z0, src, 64);
        const int cny = __shfl(ny, src, 64), cnrows = __shfl(nrows, src, 64);
        int *out = idx + ((size_t)cloud * m + (qb + (wave << 6) + src)) * nsample;
        // candidates of row r of the centre, 64 at a time: visit(j, valid) in every lane
        auto sweep_rows = [&](auto &&visit) {
          for (int r = 0; r < cnrows; ++r) {
            const int rz = r / cny, ry = r - rz * cny;
            const int cb = ((cz0 + rz) * g[1] + (cy0 + ry)) * g[0];
            const int jb = cstart[cb + cx0], je = cstart[cb + cx1 + 1];
            for (int j0 = jb; j0 < je; j0 += 64) visit(min(j0 + lane, n - 1), j0 + lane < je);
          }
        };
        int total = 0;
        sweep_rows([&](int j, bool valid) {
          const float4 c4 = sp4[j];
          const float d = rs_sqdist_expanded(cqx, cqy, cqz, cqq, c4.x, c4.y, c4.z, c4.w);
          const bool hit = valid && !(d > radius2);
          const unsigned long long mask = __ballot(hit);
          const int slot = total + rs_mbcnt(mask);
          if (hit && slot < 64) mine[slot] = sid[j];
          total += __popcll(mask);
        });
        if (total <= 64) {
          const int v = lane < total ? mine[lane] : 0x7fffffff;
          int rank = 0;
          for (int k = 0; k < total; ++k) rank += mine[k] < v ? 1 : 0;       // (broadcast reads; indices are distinct)
          if (lane < total && rank < nsample) out[rank] = v;
          const int first = (int)rs_wave_min_u32((unsigned)v);
          for (int s2 = total + lane; s2 < nsample; s2 += 64) out[s2] = first;
        } else {
          int last = -1;
          for (int s2 = 0; s2 < nsample; ++s2) {
            int best = 0x7fffffff;
            sweep_rows([&](int j, bool valid) {
              const int p = sid[j];
              if (valid && p > last && p < best) {
                const float4 c4 = sp4[j];
                const float d = rs_sqdist_expanded(cqx, cqy, cqz, cqq, c4.x, c4.y, c4.z, c4.w);
                if (!(d > radius2)) best = p;
              }
            });
            best = (int)rs_wave_min_u32((unsigned)best);
            if (lane == 0) out[s2] = best;
            last = best;
          }
        }
      }
    }
    __syncthreads();
  }
}


// ---- cell list, third design (round 5) ----------------------------------------------------------------------------
// What bounds the candidate walk (tools/probes/lds_gather.hip, profiles/r05/lds_gather.txt): 64 lanes gathering 16 random bytes
// each cost the LDS pipe 13.1 clocks per ds_read_b128 (4.8 when the addresses are consecutive), the same bytes as four
// ds_read2_b32 of a struct-of-arrays layout 57, and a ds_write_b16 per lane 7 -- a walk is 72 candidate slots per centre, so the
// gathers alone are ~940 LDS clocks per wave and 25 us per launch of 2 048 clouds, next to a VALU stream of the same order.
// ballquery_cells_kernel (rounds 2-3) pays ~11 VALU instructions per slot in a `while (ballot)` loop per row of cells and a
// branchy hit path; this kernel keeps its data structure (cloud counting-sorted by cell into one float4 array, one thread per
// centre, <= 9 contiguous candidate ranges per centre) and rebuilds the instruction stream around those costs:
//   * the walk is STRAIGHT-LINE: every row takes eight candidate slots unconditionally (a wave's longest row holds 7.7 candidates
//     on average, so the loop never ran fewer trips anyway), each ONE ds_read_b128 at an immediate offset from the row's address,
//     five VALU for the reference's distance, and TWO for the outcome: v_cmp_nlt_f32 into VCC and v_addc_co_u32 acc, acc, acc --
//     the hit bit is shifted into a per-lane word (three words of 3 rows x 8 slots); no LDS write, no branch, no exec juggling,
//     the validity of a slot (inside the row's range) is ONE mask per row applied after the walk;
//   * the 24 % of wave-rows with more than eight candidates continue in a ballot loop that appends hits directly;
//   * the hits (4.9 per centre) are decoded from the words afterwards -- find-first-bit, position, point index -- into the
//     thread's LDS row, sorted by an 8 / 12 / 16-input network chosen per wave (12 inputs: 39 compare-exchanges) and leave
//     through the coalesced write-out, which reads a row as two dwords;
//   * the build ranks a point inside its cell with the histogram's own atomic (ds_add_rtn), so the scatter needs no second
//     atomic pass; bounding-box reductions are DPP / permlane-swap butterflies instead of ds_bpermute chains; the first
//     centre of every thread is fetched before the build starts.
// Same distance arithmetic, threshold and tie rules as the scan kernel: bit-identical rows (tests/test_geometry_gpu.py).
constexpr int B3_THREADS = 512;
// HCAP: hits a thread keeps; rows with more are redone by their wave.  16 (three workgroups per CU, 80 VGPRs) or 12 (0.2 % of the
// rows of a uniform cloud overflow; 38 KB of LDS at n = 1024 and 64 VGPRs: FOUR workgroups per CU, so that 2 048 clouds are two
// full rounds of the chip instead of 2.67).  A row: HCAP hits, one overflow slot, the count = HCAP + 2 u16 (36 / 28 B: 9 / 7 dwords, odd).
constexpr int B3_SLACK = 8;                               // float4 entries behind the sorted cloud the unconditional slots may read
constexpr int B3_MAXG = 11;                               // cells per axis (LDS: three workgroups per CU at n = 1024)
constexpr int B3_MAXCELLS = B3_MAXG * B3_MAXG * B3_MAXG;

// min / max over the 64 lanes, result in every lane: four DPP steps inside the rows of 16 (the permutation and the min in ONE
// instruction: v_min_f32_dpp), then the gfx950 permlane swaps across rows.  Written as instructions: fminf / fmaxf through the
// compiler cost a separate v_mov_dpp and a canonicalising v_max per step -- 35 VALU per butterfly, 210 of the build's 479 per wave
// for the six bounding-box reductions (profiles/r05/ballquery_pmc.txt); this form is 10.  (s_nop 1: a VALU result needs two wait
// states before a DPP instruction may read it, and the assembler does not insert them inside inline assembly.)  Inputs are finite.
#define BQ_DPP4(OP)                                                                                                       \
  float r;                                                                                                                \
  asm("s_nop 1\n\t" OP "_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));            \
  asm("s_nop 1\n\t" OP "_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(r));            \
  asm("s_nop 1\n\t" OP "_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));                \
  asm("s_nop 1\n\t" OP "_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=v"(x) : "v"(r));      /* (trailing nop: the permlane swap below reads x, and the hazard pass does not see into the assembly) */
__device__ __forceinline__ float bq_wave_fmin(float x) {
  BQ_DPP4("v_min_f32")
  unsigned v = __float_as_uint(x);
  auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  float m;
  asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(__uint_as_float((unsigned)p[0])), "v"(__uint_as_float((unsigned)p[1])));
  v = __float_as_uint(m);
  auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(__uint_as_float((unsigned)q[0])), "v"(__uint_as_float((unsigned)q[1])));
  return m;
}
__device__ __forceinline__ float bq_wave_fmax(float x) {
  BQ_DPP4("v_max_f32")
  unsigned v = __float_as_uint(x);
  auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  float m;
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(__uint_as_float((unsigned)p[0])), "v"(__uint_as_float((unsigned)p[1])));
  v = __float_as_uint(m);
  auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(__uint_as_float((unsigned)q[0])), "v"(__uint_as_float((unsigned)q[1])));
  return m;
}
#undef BQ_DPP4

__device__ __forceinline__ void b3_sort12(int (&v)[16]) {
  constexpr unsigned char P12[39][2] = {{0,8},{1,7},{2,6},{3,11},{4,10},{5,9},{0,1},{2,5},{3,4},{6,9},{7,8},{10,11},{0,2},{1,6},{5,10},{9,11},
    {0,3},{1,2},{4,6},{5,7},{8,11},{9,10},{1,4},{3,5},{6,8},{7,10},{1,3},{2,5},{6,9},{8,10},{2,3},{4,5},{6,7},{8,9},{4,6},{5,7},{3,4},{5,6},{7,8}};
#pragma unroll
  for (int i = 0; i < 39; ++i) bc_cx(v[P12[i][0]], v[P12[i][1]]);
}

// acc = 2 * acc + (not (d > r2)): the outcome of one candidate slot, two VALU instructions
__device__ __forceinline__ void b3_record(unsigned &acc, float d, float r2) {
  asm("v_cmp_nlt_f32_e64 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(d), "s"(r2) : "vcc");
}

// PHASE (round 6): 0 = build + query in one launch (above); 1 = BUILD only -- the cloud's cell list leaves LDS for a per-cloud image in
// HBM (rs_ballquery_grid_build); 2 = QUERY only -- the image comes back into LDS as it lay there, a straight copy, and the centres
// are walked (rs_ballquery_grid_query).  One build serves every query on the same coordinates and radius; the per-launch build was 380
// of the 1 976 VALU instructions per wave and 18.6 of 66 us at 2 048 clouds (profiles/r05/ballquery_pmc.txt).
// Image of one cloud (B3_HEADER bytes of header, then the LDS bytes [sp4 | sid | cstart]):
//   header: float lo[3], inv; int g[3], ncell
constexpr int B3_HEADER = 32;
__host__ __device__ constexpr long long b3_image_bytes(int n) {
  return (long long)(n + B3_SLACK) * 16 + (long long)((n + 1) & ~1) * 2 + (long long)sizeof(int) * (B3_MAXCELLS + 1);
}
__host__ __device__ constexpr long long b3_image_stride(int n) { return (B3_HEADER + b3_image_bytes(n) + 15) & ~15LL; }

template <int PP, int B3_HCAP, int B3_WAVES, int PHASE = 0>
__global__ void __launch_bounds__(B3_THREADS, B3_WAVES)
ballquery_cells3_kernel(int b, int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
                        const float *__restrict__ xyz, int *__restrict__ idx, int *__restrict__ cnt_out, int dbg, char *__restrict__ image = nullptr) {
  constexpr int B3_RSTRIDE = B3_HCAP + 2, B3_COUNT = B3_RSTRIDE - 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4 *sp4 = reinterpret_cast<float4 *>(lds);                       // (x, y, z, |p|^2) sorted by cell (+ B3_SLACK entries)
  unsigned short *sid = reinterpret_cast<unsigned short *>(lds + 4 * (n + B3_SLACK));
  int *cstart = reinterpret_cast<int *>(sid + ((n + 1) & ~1));          // counts, then running starts (ncell + 1)
  unsigned short *rows = reinterpret_cast<unsigned short *>(cstart + B3_MAXCELLS + 1);     // one row of B3_RSTRIDE u16 per thread
  __shared__ float red[6][B3_THREADS / 64];
  __shared__ int wsum[B3_THREADS / 64];

  const int cloud = blockIdx.x;
  const float *pts = xyz + (size_t)cloud * n * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float r2u = rs_uniform(radius2);

  // the first centre of this thread: in flight while the grid is built (or copied in)
  float cq0 = 0.f, cq1 = 0.f, cq2 = 0.f;
  if (PHASE != 1 && tid < m) { const float *c = new_xyz + ((size_t)cloud * m + tid) * 3; cq0 = c[0]; cq1 = c[1]; cq2 = c[2]; }

  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float inv = 0.f;
  int g[3] = {1, 1, 1};
  char *img = image ? image + (size_t)cloud * (size_t)b3_image_stride(n) : nullptr;
  if constexpr (PHASE == 2) {
    // the image as it lay in LDS when the build left it: 16-byte units, consecutive lanes consecutive addresses
    const int units = (int)(b3_image_bytes(n) >> 4), tail = (int)(b3_image_bytes(n) & 15) >> 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(img + B3_HEADER);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (int u = tid; u < units; u += B3_THREADS) dst[u] = src[u];
    if (tid < tail) reinterpret_cast<unsigned *>(lds)[units * 4 + tid] = reinterpret_cast<const unsigned *>(img + B3_HEADER)[units * 4 + tid];
    const float4 h0 = *reinterpret_cast<const float4 *>(img);
    const int4 h1 = *reinterpret_cast<const int4 *>(img + 16);
    lo[0] = rs_uniform(h0.x); lo[1] = rs_uniform(h0.y); lo[2] = rs_uniform(h0.z); inv = rs_uniform(h0.w);
    g[0] = __builtin_amdgcn_readfirstlane(h1.x); g[1] = __builtin_amdgcn_readfirstlane(h1.y); g[2] = __builtin_amdgcn_readfirstlane(h1.z);
    __syncthreads();
  } else {
  // A. this thread's points (registers), bounding box; counts zeroed
  float px[PP], py[PP], pz[PP];
#pragma unroll
  for (int k = 0; k < PP; ++k) {
    const int p = tid + k * B3_THREADS;
    if (p < n) {
      px[k] = pts[p * 3 + 0]; py[k] = pts[p * 3 + 1]; pz[k] = pts[p * 3 + 2];
      lo[0] = fminf(lo[0], px[k]); lo[1] = fminf(lo[1], py[k]); lo[2] = fminf(lo[2], pz[k]);
      hi[0] = fmaxf(hi[0], px[k]); hi[1] = fmaxf(hi[1], py[k]); hi[2] = fmaxf(hi[2], pz[k]);
    } else { px[k] = py[k] = pz[k] = 0.f; }
  }
  for (int c = tid; c <= B3_MAXCELLS; c += B3_THREADS) cstart[c] = 0;
  if (tid < B3_SLACK) sp4[n + tid] = make_float4(0.f, 0.f, 0.f, INFINITY);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float l = bq_wave_fmin(lo[a]), h = bq_wave_fmax(hi[a]);
    if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
  }
  __syncthreads();
  float ext = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = red[a][0], h = red[3 + a][0];
#pragma unroll
    for (int w = 1; w < B3_THREADS / 64; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
    lo[a] = l; hi[a] = h;
    ext = fmaxf(ext, h - l);
  }
  // B. grid geometry (cell edge >= 1.001 r: a point inside the computed radius lies in the 27 cells around the centre's)
  const float cell = fmaxf(sqrtf(radius2) * 1.001f, ext * (1.0001f / B3_MAXG));
  inv = 1.0f / cell;
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = min(B3_MAXG, (int)((hi[a] - lo[a]) * inv) + 1);
  const int ncell = g[0] * g[1] * g[2];

  // C. histogram; the atomic's return value is the point's rank inside its cell
  int pc[PP], rk[PP];
#pragma unroll
  for (int k = 0; k < PP; ++k)
    if (tid + k * B3_THREADS < n) {
      const int cx = min(g[0] - 1, (int)((px[k] - lo[0]) * inv));
      const int cy = min(g[1] - 1, (int)((py[k] - lo[1]) * inv));
      const int cz = min(g[2] - 1, (int)((pz[k] - lo[2]) * inv));
      pc[k] = (cz * g[1] + cy) * g[0] + cx;
      rk[k] = atomicAdd(&cstart[pc[k]], 1);
    }
  __syncthreads();
  // D. exclusive scan of the counts: a run of consecutive cells per thread, wave scan, one LDS hop
  const int per = (ncell + B3_THREADS - 1) / B3_THREADS;
  int run = 0;
  for (int k = 0; k < per; ++k) { const int c = tid * per + k; if (c < ncell) run += cstart[c]; }
  int incl = run;
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - run;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int k = 0; k < per; ++k) {
    const int c = tid * per + k;
    if (c < ncell) { const int v = cstart[c]; cstart[c] = base; base += v; }
  }
  if (tid == 0) cstart[ncell] = n;
  __syncthreads();
  // E. place the points (order inside a cell is arbitrary: hits are sorted by index afterwards)
#pragma unroll
  for (int k = 0; k < PP; ++k) {
    const int p = tid + k * B3_THREADS;
    if (p < n) {
      const int pos = cstart[pc[k]] + rk[k];
      sp4[pos] = make_float4(px[k], py[k], pz[k], rs_sqnorm(px[k], py[k], pz[k]));
      sid[pos] = (unsigned short)p;
    }
  }
  __syncthreads();
  if constexpr (PHASE == 1) {
    // the cell list leaves for the cloud's image: [sp4 | sid | cstart] as it lies here (the query copies it straight back)
    const int units = (int)(b3_image_bytes(n) >> 4), tail = (int)(b3_image_bytes(n) & 15) >> 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(lds);
    uint4 *dst = reinterpret_cast<uint4 *>(img + B3_HEADER);
    for (int u = tid; u < units; u += B3_THREADS) dst[u] = src[u];
    if (tid < tail) reinterpret_cast<unsigned *>(img + B3_HEADER)[units * 4 + tid] = reinterpret_cast<const unsigned *>(lds)[units * 4 + tid];
    if (tid == 0) {
      *reinterpret_cast<float4 *>(img) = make_float4(lo[0], lo[1], lo[2], inv);
      *reinterpret_cast<int4 *>(img + 16) = make_int4(g[0], g[1], g[2], ncell);
    }
    return;
  }
  if (dbg == 1) return;                                         // (experiment: cost of the build alone)
  }

  // F. centres, B3_THREADS at a time
  unsigned short *row = rows + tid * B3_RSTRIDE;
  for (int qb = 0; qb < m; qb += B3_THREADS) {
    const int q = qb + tid;
    const bool qv = q < m;
    float qx = cq0, qy = cq1, qz = cq2;
    if (qb > 0 && qv) { const float *c = new_xyz + ((size_t)cloud * m + q) * 3; qx = c[0]; qy = c[1]; qz = c[2]; }
    const float qq = rs_sqnorm(qx, qy, qz);
    int x0 = 1, x1 = 0, y0 = 0, z0 = 0, ny = 0, nz = 0, nrows = 0;
    if (qv) {
      const float fx = fminf(fmaxf((qx - lo[0]) * inv, -2.f), (float)B3_MAXG + 2.f);
      const float fy = fminf(fmaxf((qy - lo[1]) * inv, -2.f), (float)B3_MAXG + 2.f);
      const float fz = fminf(fmaxf((qz - lo[2]) * inv, -2.f), (float)B3_MAXG + 2.f);
      const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
      x0 = max(cx - 1, 0); x1 = min(cx + 1, g[0] - 1);
      y0 = max(cy - 1, 0); const int y1 = min(cy + 1, g[1] - 1);
      z0 = max(cz - 1, 0); const int z1 = min(cz + 1, g[2] - 1);
      ny = max(y1 - y0 + 1, 0); nz = max(z1 - z0 + 1, 0);
      if (x0 > x1) ny = 0;                                        // (a centre more than a cell outside the box: no row)
      nrows = ny * nz;
    }
    unsigned jl[9];                                             // a row's range: first position | length << 16 (one register per row)
    {
      const int cb0 = (z0 * g[1] + y0) * g[0], xa = x0, xb = x1 + 1;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int rz = r / 3, ry = r - rz * 3;                   // (ry, rz) enumerate a 3 x 3 block; clipped rows are empty
        const bool on = ry < ny && rz < nz;
        const int cb = cb0 + (rz * g[1] + ry) * g[0];
        const int j0 = cstart[on ? cb + xa : 0], j1 = cstart[on ? cb + xb : 0];       // (off: twice the same entry, length 0)
        jl[r] = (unsigned)j0 | ((unsigned)(j1 - j0) << 16);
      }
    }
    // The walk: row r = 3 w + k puts its eight slots into word w; after the walk slot s of that row sits at bit 23 - (8 k + s).
    int cnt = 0;                                                // hits appended so far (the true count; the row keeps <= HCAP + 1)
    unsigned acc[3] = {0u, 0u, 0u}, valid[3] = {0u, 0u, 0u};
    // (which row goes into which word: the centre's own row of cells holds ~1.5 of a centre's 4.9 hits, the four edge rows ~0.6 each, the
    //  corners ~0.25 -- in natural order the middle word carries 2.7 of them and the decode loop, which runs as long as the busiest word of
    //  the busiest lane has bits, pays for it; interleaved {4,0,2} {1,3,6} {5,7,8} the words expect 2.0 / 1.45 / 1.45)
    constexpr int B3_ROW[9] = {4, 0, 2, 1, 3, 6, 5, 7, 8};
#pragma unroll
    for (int ri = 0; ri < 9; ++ri) {
      const int r = ri;                                           // position in the words; the row of cells it holds is B3_ROW[ri]
      const unsigned jlr = jl[B3_ROW[ri]];
      const int jbr = (int)(jlr & 0xffffu), lnr = (int)(jlr >> 16);
      const float4 *P = sp4 + jbr;
#pragma unroll
      for (int h = 0; h < 8; h += 4) {                            // four gathers in flight together (eight would spill)
        float4 c4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) c4[s] = P[h + s];
#pragma unroll
        for (int s = 0; s < 4; ++s)
          b3_record(acc[r / 3], rs_sqdist_expanded(qx, qy, qz, qq, c4[s].x, c4[s].y, c4[s].z, c4[s].w), r2u);
      }
      valid[r / 3] |= ((0xff00u >> min(lnr, 8)) & 0xffu) << (8 * (2 - r % 3));      // the first min(len, 8) slots: the top bits of the row's byte
      if (__ballot(lnr > 8)) {                                    // ~ a quarter of the wave-rows: the longest row of 64 lanes holds 7.7
        for (int s = 8; __ballot(s < lnr); ++s) {
          const int j = min(jbr + s, n);
          const float4 c4 = sp4[j];
          const float d = rs_sqdist_expanded(qx, qy, qz, qq, c4.x, c4.y, c4.z, c4.w);
          if (s < lnr && !(d > radius2)) {
            if (cnt <= B3_HCAP) row[cnt] = sid[j];
            ++cnt;
          }
        }
      }
    }
    if (dbg == 2) { if ((acc[0] ^ acc[1] ^ acc[2]) == 0x12345u) idx[0] = cnt; continue; }      // (experiment: build + walk)
    // Hits out of the words, the three words side by side (independent find-first-bit / lookup chains in one loop): highest bit
    // first = ascending slot; position -> point index -> the thread's row.
    {
      unsigned a0 = acc[0] & valid[0], a1 = acc[1] & valid[1], a2 = acc[2] & valid[2];
      auto take = [&](unsigned &a, unsigned ra, unsigned rb, unsigned rc) {      // (the three rows of the word, by value: no indexed array)
        if (a != 0u) {
          const int bit = 31 - __clz((int)a);
          a &= ~(1u << bit);
          const unsigned jr = bit >= 16 ? ra : (bit >= 8 ? rb : rc);
          const int j = (int)(jr & 0xffffu) + ((23 - bit) & 7);       // slot s of row k sits at bit 23 - (8 k + s)
          const int sv = sid[j];
          if (cnt <= B3_HCAP) row[cnt] = (unsigned short)sv;
          ++cnt;
        }
      };
      const unsigned r0 = jl[B3_ROW[0]], r1 = jl[B3_ROW[1]], r2_ = jl[B3_ROW[2]], r3 = jl[B3_ROW[3]], r4 = jl[B3_ROW[4]], r5 = jl[B3_ROW[5]],
                     r6 = jl[B3_ROW[6]], r7 = jl[B3_ROW[7]], r8 = jl[B3_ROW[8]];
      while (__ballot((a0 | a1 | a2) != 0u)) { take(a0, r0, r1, r2_); take(a1, r3, r4, r5); take(a2, r6, r7, r8); }
    }
    const bool over = cnt > B3_HCAP;
    if (qv && cnt_out && !over) cnt_out[(size_t)cloud * m + q] = cnt > 0 ? min(cnt, nsample) : 1;
    // sorted in registers (wave-uniform choice of the network: 8, 12 or 16 inputs)
    {
      const int kept = over ? 0 : cnt;
      const int wmax = __builtin_amdgcn_readfirstlane((int)rs_wave_max_u32((unsigned)kept));
      auto sorted = [&](auto width) {
        constexpr int W = decltype(width)::value;
        int v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < W; ++i) { const int e = row[i]; v[i] = i < kept ? e : 0x7fffffff; }
        if constexpr (W == 8) bc_sort<8>(v); else if constexpr (W == 12) b3_sort12(v); else bc_sort<16>(v);
        unsigned *row32 = reinterpret_cast<unsigned *>(row);
#pragma unroll
        for (int i = 0; i < W; i += 2) row32[i >> 1] = ((unsigned)v[i] & 0xffffu) | ((unsigned)v[i + 1] << 16);
      };
      if constexpr (B3_HCAP > 12) { if (wmax > 12) sorted(std::integral_constant<int, 16>{}); }
      if (wmax > 8 && wmax <= 12) sorted(std::integral_constant<int, 12>{});
      else if (wmax > 1 && wmax <= 8) sorted(std::integral_constant<int, 8>{});
    }
    row[B3_COUNT] = over ? (unsigned short)0xffffu : (unsigned short)cnt;
    __syncthreads();
    if (dbg == 3) { __syncthreads(); continue; }                  // (experiment: build + walk + sort)
    // coalesced write-out of the rows with <= B3_HCAP hits; padding = the lowest hit (an empty ball yields zeros)
    const int nq = min(B3_THREADS, m - qb);
    int *dst = idx + ((size_t)cloud * m + qb) * nsample;
    if ((nsample & 3) == 0) {
      // a row is gpr = nsample / 4 groups of four slots, one 16-byte store each.  The group a thread writes and the distance between
      // its rows are the same in every trip (B3_THREADS is a multiple of gpr for the usual nsample = 8 .. 128; otherwise they advance
      // by a fixed step): nothing but the row's count / lowest hit / two dwords of hits is fetched inside the loop.
      const int gpr = nsample >> 2;
      int ql = tid / gpr, g4 = (tid - ql * gpr) << 2;
      const int dq = B3_THREADS / gpr, dg = (B3_THREADS - dq * gpr) << 2;
      int4 *dst4 = reinterpret_cast<int4 *>(dst);
      for (int e = tid; e < nq * gpr; e += B3_THREADS) {
        const unsigned short *h = rows + ql * B3_RSTRIDE;
        const int c = h[B3_COUNT];
        if (c != 0xffff) {
          const int take = min(c, nsample);
          const int first = c > 0 ? h[0] : 0;
          int4 v = make_int4(first, first, first, first);
          if (g4 < take) {                                        // (then g4 + 3 < B3_HCAP + 2: inside the row)
            const unsigned *h32 = reinterpret_cast<const unsigned *>(h);
            const unsigned a = h32[g4 >> 1], bb = h32[(g4 >> 1) + 1];
            v.x = (int)(a & 0xffffu);
            v.y = g4 + 1 < take ? (int)(a >> 16) : first;
            v.z = g4 + 2 < take ? (int)(bb & 0xffffu) : first;
            v.w = g4 + 3 < take ? (int)(bb >> 16) : first;
          }
          dst4[e] = v;
        }
        ql += dq; g4 += dg;
        if (g4 >= nsample) { g4 -= nsample; ++ql; }
      }
    } else {
      int ql = tid / nsample, sl = tid - ql * nsample;
      const int dq = B3_THREADS / nsample, ds = B3_THREADS - dq * nsample;
      for (int e = tid; e < nq * nsample; e += B3_THREADS) {
        const unsigned short *h = rows + ql * B3_RSTRIDE;
        const int c = h[B3_COUNT];
        if (c != 0xffff) {
          const int take = min(c, nsample);
          dst[e] = c > 0 ? h[sl < take ? sl : 0] : 0;
        }
        ql += dq; sl += ds;
        if (sl >= nsample) { sl -= nsample; ++ql; }
      }
    }
    // Rows with more than B3_HCAP hits (dense clusters): the WAVE redoes them together, one row at a time -- 64 lanes over the
    // row's candidates, hits through a ballot into the wave's arena, ranks by counting (<= 64 hits) or the nsample lowest
    // indices by repeated minimum (more).
    __syncthreads();                                              // (the write-out has read every row: a wave's own rows now serve as its arena)
    {
      unsigned long long todo = __ballot(qv && over);
      int *mine = reinterpret_cast<int *>(rows + (wave * 64) * B3_RSTRIDE);      // 64 ints inside 64 rows of >= 28 B
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const float cqx = __shfl(qx, src, 64), cqy = __shfl(qy, src, 64), cqz = __shfl(qz, src, 64), cqq = __shfl(qq, src, 64);
        const int cx0 = __shfl(x0, src, 64), cx1 = __shfl(x1, src, 64), cy0 = __shfl(y0, src, 64), cz0 = __shfl(z0, src, 64);
        const int cny = __shfl(ny, src, 64), cnrows = __shfl(nrows, src, 64);
        const size_t qrow = (size_t)cloud * m + (qb + (wave << 6) + src);
        int *out = idx + qrow * nsample;
        auto sweep_rows = [&](auto &&visit) {
          for (int r = 0; r < cnrows; ++r) {
            const int rz = r / cny, ry = r - rz * cny;
            const int cb = ((cz0 + rz) * g[1] + (cy0 + ry)) * g[0];
            const int b0 = cstart[cb + cx0], e0 = cstart[cb + cx1 + 1];
            for (int j0 = b0; j0 < e0; j0 += 64) visit(min(j0 + lane, n - 1), j0 + lane < e0);
          }
        };
        int total = 0;
        sweep_rows([&](int j, bool valid_) {
          const float4 c4 = sp4[j];
          const float d = rs_sqdist_expanded(cqx, cqy, cqz, cqq, c4.x, c4.y, c4.z, c4.w);
          const bool hit = valid_ && !(d > radius2);
          const unsigned long long mask = __ballot(hit);
          const int slot = total + rs_mbcnt(mask);
          if (hit && slot < 64) mine[slot] = sid[j];
          total += __popcll(mask);
        });
        if (cnt_out && lane == 0) cnt_out[qrow] = min(total, nsample);
        if (total <= 64) {
          const int v = lane < total ? mine[lane] : 0x7fffffff;
          int rank = 0;
          for (int k = 0; k < total; ++k) rank += mine[k] < v ? 1 : 0;       // (broadcast reads; indices are distinct)
          if (lane < total && rank < nsample) out[rank] = v;
          const int first = (int)rs_wave_min_u32((unsigned)v);
          for (int s2 = total + lane; s2 < nsample; s2 += 64) out[s2] = first;
        } else {
          int last = -1;
          for (int s2 = 0; s2 < nsample; ++s2) {
            int best = 0x7fffffff;
            sweep_rows([&](int j, bool valid_) {
              const int p = sid[j];
              if (valid_ && p > last && p < best) {
                const float4 c4 = sp4[j];
                const float d = rs_sqdist_expanded(cqx, cqy, cqz, cqq, c4.x, c4.y, c4.z, c4.w);
                if (!(d > radius2)) best = p;
              }
            });
            best = (int)rs_wave_min_u32((unsigned)best);
            if (lane == 0) out[s2] = best;
            last = best;
          }
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

// ---- the cell list as a reusable per-cloud image (round 6) ----------------------------------------------------------------------------
extern "C" long long rs_ballquery_grid_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  return (long long)b * b3_image_stride(n);
}

static bool b3_grid_shape_ok(int n, int nsample) { return n >= 64 && n <= 8 * B3_THREADS && n <= BG_MAXN && nsample >= 1 && nsample <= 64; }

extern "C" int rs_ballquery_grid_build(int b, int n, float radius2, const float *xyz, void *image, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0, "rs_ballquery_grid_build: negative size");
  if (b == 0) return RS_OK;
  RS_REQUIRE(xyz && image && ((uintptr_t)image & 15) == 0, "rs_ballquery_grid_build: null or unaligned pointer");
  RS_REQUIRE(b3_grid_shape_ok(n, 1) && radius2 > 0.f, "rs_ballquery_grid_build: clouds of 64 .. %d points and a positive radius (n=%d)", 8 * B3_THREADS, n);
  const size_t lds = (size_t)b3_image_bytes(n) + (size_t)B3_THREADS * (16 + 2) * 2;
  RS_REQUIRE_LDS(lds, "rs_ballquery_grid_build");
  hipStream_t st = (hipStream_t)stream;
#define RS_B3_BUILD(PP) hipLaunchKernelGGL((ballquery_cells3_kernel<PP, 16, 6, 1>), dim3(b), dim3(B3_THREADS), lds, st, b, n, 0, radius2, 0, nullptr, xyz, nullptr, nullptr, 0, (char *)image)
  if (n <= 2 * B3_THREADS) RS_B3_BUILD(2); else if (n <= 4 * B3_THREADS) RS_B3_BUILD(4); else RS_B3_BUILD(8);
#undef RS_B3_BUILD
  RS_CHECK_LAUNCH("rs_ballquery_grid_build");
  return RS_OK;
}

extern "C" int rs_ballquery_grid_query(int b, int n, int m, float radius2, int nsample, const float *new_xyz, const void *image,
                                       int *idx, int *cnt, void *stream) {
  RS_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "rs_ballquery_grid_query: negative size");
  if (b == 0 || m == 0 || nsample == 0) return RS_OK;
  RS_REQUIRE(new_xyz && image && idx && ((uintptr_t)image & 15) == 0, "rs_ballquery_grid_query: null or unaligned pointer");
  RS_REQUIRE(b3_grid_shape_ok(n, nsample), "rs_ballquery_grid_query: clouds of 64 .. %d points, nsample 1 .. 64 (n=%d, nsample=%d)", 8 * B3_THREADS, n, nsample);
  const size_t lds = (size_t)b3_image_bytes(n) + (size_t)B3_THREADS * (16 + 2) * 2;
  RS_REQUIRE_LDS(lds, "rs_ballquery_grid_query");
  hipLaunchKernelGGL((ballquery_cells3_kernel<2, 16, 6, 2>), dim3(b), dim3(B3_THREADS), lds, (hipStream_t)stream, b, n, m, radius2, nsample, new_xyz,
                     nullptr, idx, cnt, 0, (char *)image);
  RS_CHECK_LAUNCH("rs_ballquery_grid_query");
  return RS_OK;
}

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
  // 2 (default where the grid applies): the straight-line pair walk (round 5); 1: the register-carried, cell-sorted variant with
  // the ballot loop (rounds 2-3); 0: the first cell-list kernel
  static const int cells_on = getenv("RS_BALLQUERY_CELLS") ? atoi(getenv("RS_BALLQUERY_CELLS")) : 2;
  if (use_grid && grid_ok && cells_on >= 2 && nsample >= 1 && nsample <= 256 && n <= 8 * B3_THREADS) {
    // round 5: the straight-line pair walk (ballquery_cells3_kernel); points a thread carries through the build: 2 / 4 / 8
    static const int dbg = getenv("RS_BALLQUERY_DBG") ? atoi(getenv("RS_BALLQUERY_DBG")) : 0;
    // RS_BALLQUERY_OCC: 6 (default) = 16 kept hits, 80 VGPRs, three workgroups per CU at n = 1024; 8 = 12 kept hits, 64 VGPRs, four -- measured
    // SLOWER (85.9 against 71.7 us at 2 048 clouds: 76 B of scratch per lane outside the walk, profiles/r05/ballquery_occupancy_ab.txt)
    static const int occ = getenv("RS_BALLQUERY_OCC") ? atoi(getenv("RS_BALLQUERY_OCC")) : 6;
    const int hcap = occ >= 8 ? 12 : 16;
    const size_t lds = (size_t)(n + B3_SLACK) * 16 + (size_t)((n + 1) & ~1) * 2 + sizeof(int) * (B3_MAXCELLS + 1) +
                       (size_t)B3_THREADS * (hcap + 2) * 2;
    RS_REQUIRE_LDS(lds, "rs_ballquery");
#define RS_B3_LAUNCH(PP, HC, WV) hipLaunchKernelGGL((ballquery_cells3_kernel<PP, HC, WV>), dim3(b), dim3(B3_THREADS), lds, st, b, n, m, radius2, nsample, new_xyz, xyz, idx, cnt, dbg)
    if (hcap == 12) {
      if (n <= 2 * B3_THREADS) RS_B3_LAUNCH(2, 12, 8); else if (n <= 4 * B3_THREADS) RS_B3_LAUNCH(4, 12, 8); else RS_B3_LAUNCH(8, 12, 8);
    } else {
      if (n <= 2 * B3_THREADS) RS_B3_LAUNCH(2, 16, 6); else if (n <= 4 * B3_THREADS) RS_B3_LAUNCH(4, 16, 6); else RS_B3_LAUNCH(8, 16, 6);
    }
#undef RS_B3_LAUNCH
    RS_CHECK_LAUNCH("rs_ballquery");
    return RS_OK;
  }
  if (use_grid && grid_ok && cells_on && nsample >= 1 && nsample <= 256) {
    static const int dbg = getenv("RS_BALLQUERY_DBG") ? atoi(getenv("RS_BALLQUERY_DBG")) : 0;
    // threads per cloud: 512 (256 = two passes over the centres with half the waves: 154 against 114 us, profiles/r03/)
    static const int threads = getenv("RS_BALLQUERY_THREADS") ? atoi(getenv("RS_BALLQUERY_THREADS")) : 512;
    const int t = threads == 512 ? 512 : 256;
    const size_t hit_bytes = (size_t)t * (BC_HCAP + 1) * 2, cursor_bytes = sizeof(int) * BG_MAXCELLS;
    const size_t lds = (size_t)(n + BC_SLACK) * 16 + (size_t)n * 4 + sizeof(int) * (BG_MAXCELLS + 1 + 2 * (size_t)t) +
                       (hit_bytes > cursor_bytes ? hit_bytes : cursor_bytes);
    // 6 waves per SIMD (<= 80 VGPRs) lets three 512-thread workgroups share a CU when their LDS (45 KB at n = 1024) allows it
    static const int occ = getenv("RS_BALLQUERY_OCC") ? atoi(getenv("RS_BALLQUERY_OCC")) : 6;
    if (t == 512 && occ >= 6)
      hipLaunchKernelGGL((ballquery_cells_kernel<512, 6>), dim3(b), dim3(512), lds, st, b, n, m, radius2, nsample, new_xyz, xyz, idx, cnt, dbg);
    else if (t == 512)
      hipLaunchKernelGGL((ballquery_cells_kernel<512, 4>), dim3(b), dim3(512), lds, st, b, n, m, radius2, nsample, new_xyz, xyz, idx, cnt, dbg);
    else
      hipLaunchKernelGGL((ballquery_cells_kernel<256, 4>), dim3(b), dim3(256), lds, st, b, n, m, radius2, nsample, new_xyz, xyz, idx, cnt, dbg);
    RS_CHECK_LAUNCH("rs_ballquery");
    return RS_OK;
  }
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
