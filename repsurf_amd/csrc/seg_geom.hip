// seg_geom.hip — packed-batch ("offset") geometry of the segmentation path for gfx950.
//
// Layout: clouds are concatenated, xyz (Ntot, 3); offset (B) int32 holds the running row ends, cloud i owns
// rows [offset[i-1], offset[i]) (segmentation/util/data_util.py:15-23 collate format).
//
// rs_knnquery_offset   pointops.knnquery (segmentation/modules/pointops/functions/pointops.py:114-130 ->
//                      src/knnquery/knnquery_cuda_kernel.cu:65-108): direct-difference squared distances,
//                      initial list (1e10, first row of the cloud), strict '<' replacement, ascending
//                      output.  The reference runs one thread per query that re-reads the whole cloud from
//                      global memory and keeps a 100-entry heap in scratch.  Here a workgroup stages the
//                      cloud through LDS once for 64 queries, 4 lanes share a query (each scans every 4th
//                      point, lists merged lexicographically in (distance, row) so the result equals the
//                      sequential scan), and workgroups that straddle a cloud boundary walk the (at most few)
//                      clouds their queries belong to.  Exact distance ties come out in ascending row order
//                      (the reference's heap leaves them in heap order; parity unpinned, oracle/geom_oracle.c).
// rs_umbrella_fan_offset  everything UmbrellaSurfaceConstructor.forward does between the kNN and self.mlps
//                      (segmentation/modules/repsurface_utils.py:77-98,305-321): the point itself stays in the
//                      ring (k triangles, the two that touch it are degenerate and take the first valid
//                      triangle's normal/centroid/constant), azimuth after the fixed rotation (sort='fix').
// rs_interp_weights    inverse-distance weights of SurfaceFeaturePropagationCD / pointops.interpolation
//                      (repsurface_utils.py:262-265, pointops.py:262-265).
#include "rs_common.h"
#include "umbrella_fan.h"

namespace {

constexpr int SG_THREADS = 256;
constexpr int SG_TILE = 2048;         // points per LDS tile (32 KB as float4)
constexpr int SG_LANES = 4;           // lanes per query
constexpr int SG_QPB = SG_THREADS / SG_LANES;

// first cloud c with q < ends[c]  (the reference walks linearly: knnquery_cuda_kernel.cu:51-62)
__device__ __forceinline__ int cloud_of(int q, const int *__restrict__ ends, int b) {
  int lo = 0, hi = b - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (q < ends[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

template <int K>
__device__ __forceinline__ void sorted_insert(float (&bd)[K], int (&bi)[K], float d, int p) {
  if (d < bd[K - 1]) {          // strict: an equal distance never displaces an earlier row
    bd[K - 1] = d; bi[K - 1] = p;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool sw = bd[j] < bd[j - 1];
      const float td = bd[j]; const int ti = bi[j];
      bd[j] = sw ? bd[j - 1] : td; bi[j] = sw ? bi[j - 1] : ti;
      bd[j - 1] = sw ? td : bd[j - 1]; bi[j - 1] = sw ? ti : bi[j - 1];
    }
  }
}

template <int K>
__device__ __forceinline__ void sorted_insert_lex(float (&bd)[K], int (&bi)[K], float d, int p) {
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

// LDS_MERGE = false: the 4 partial lists are merged by two butterfly rounds of shuffles (K*K compare-swaps
// per round, fine up to K = 16).  LDS_MERGE = true: the lists go to LDS and lane 0 of the query does a 4-way
// merge by head pointers (K steps) -- for K = 32/64, where the unrolled butterfly would spill.
template <int K, bool LDS_MERGE>
__global__ void __launch_bounds__(SG_THREADS)
knn_packed_kernel(int m, int nsample, int b, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                  const int *__restrict__ offset, const int *__restrict__ new_offset, int *__restrict__ idx,
                  float *__restrict__ dist2) {
  constexpr int LIST_BYTES = LDS_MERGE ? SG_THREADS * K * 8 : 0;
  constexpr int LDS_BYTES = (LIST_BYTES > SG_TILE * 16) ? LIST_BYTES : SG_TILE * 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  float4 *tile = reinterpret_cast<float4 *>(lds);

  const int sub = threadIdx.x & (SG_LANES - 1);
  const int q0 = blockIdx.x * SG_QPB;
  const int q = q0 + (threadIdx.x >> 2);
  const int qc = min(q, m - 1);
  const int mycloud = cloud_of(qc, new_offset, b);
  const int c_lo = cloud_of(q0, new_offset, b);                        // workgroup-uniform
  const int c_hi = cloud_of(min(q0 + SG_QPB - 1, m - 1), new_offset, b);
  const float qx = new_xyz[qc * 3 + 0], qy = new_xyz[qc * 3 + 1], qz = new_xyz[qc * 3 + 2];

  float bd[K]; int bi[K];
  {
    const int mystart = mycloud ? offset[mycloud - 1] : 0;
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = 1e10f; bi[j] = mystart; }     // knnquery_cuda_kernel.cu:86-87
  }
  for (int c = c_lo; c <= c_hi; ++c) {
    const int start = c ? offset[c - 1] : 0, end = offset[c];
    for (int t0 = start; t0 < end; t0 += SG_TILE) {
      const int tn = min(SG_TILE, end - t0);
      __syncthreads();
      for (int p = threadIdx.x; p < tn; p += SG_THREADS) {
        const float *s = xyz + (size_t)(t0 + p) * 3;
        tile[p] = make_float4(s[0], s[1], s[2], 0.f);
      }
      __syncthreads();
      // 8 candidates per lane and round, then one insertion per trip of a wave-uniform loop for the candidates that
      // beat the lane's K-th distance (see knn_scan4 in knn_umbrella.hip: a wave nearly always has SOME lane inserting,
      // so candidate-by-candidate the K-step insertion ran for ~90 % of the candidates).  Lanes of other clouds hold no
      // pending candidates.  Same candidates, same order per lane, same comparisons.
      constexpr int U = 8;
      for (int p0 = sub; p0 < tn; p0 += SG_LANES * U) {
        float d[U];
        unsigned pend = 0;
        if (mycloud == c) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int p = p0 + SG_LANES * u;
            const float4 v = tile[min(p, tn - 1)];
            const float dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
            d[u] = p < tn ? (dx * dx + dy * dy) + dz * dz : INFINITY;      // :93, no contraction
            pend |= (d[u] < bd[K - 1] ? 1u : 0u) << u;
          }
        }
        while (__any(pend != 0)) {
          if (pend) {
            const int u = __ffs(pend) - 1;
            pend &= pend - 1;
            float dd = d[0];
#pragma unroll
            for (int v = 1; v < U; ++v) dd = (u == v) ? d[v] : dd;
            sorted_insert<K>(bd, bi, dd, t0 + p0 + SG_LANES * u);
          }
        }
      }
    }
  }

  if (!LDS_MERGE) {
#pragma unroll
    for (int mask = 1; mask <= 2; mask <<= 1) {
      float od[K]; int oi[K];
#pragma unroll
      for (int j = 0; j < K; ++j) { od[j] = __shfl_xor(bd[j], mask, 64); oi[j] = __shfl_xor(bi[j], mask, 64); }
#pragma unroll
      for (int j = 0; j < K; ++j) sorted_insert_lex<K>(bd, bi, od[j], oi[j]);
    }
    if (q < m && sub == 0) {
#pragma unroll
      for (int j = 0; j < K; ++j) if (j < nsample) {
        idx[(size_t)q * nsample + j] = bi[j];
        if (dist2) dist2[(size_t)q * nsample + j] = bd[j];
      }
    }
  } else {
    __syncthreads();                       // the tile is dead: its LDS becomes the list store
    float *ld = reinterpret_cast<float *>(lds);
    int *li = reinterpret_cast<int *>(lds) + SG_THREADS * K;
    // entry j of thread t at [j * 256 + t]: consecutive lanes hit consecutive banks
#pragma unroll
    for (int j = 0; j < K; ++j) { ld[j * SG_THREADS + threadIdx.x] = bd[j]; li[j * SG_THREADS + threadIdx.x] = bi[j]; }
    __syncthreads();
    if (q < m && sub == 0) {
      int h[SG_LANES]; float hd[SG_LANES]; int hi_[SG_LANES];
#pragma unroll
      for (int l = 0; l < SG_LANES; ++l) { h[l] = 0; hd[l] = ld[threadIdx.x + l]; hi_[l] = li[threadIdx.x + l]; }
      for (int j = 0; j < nsample; ++j) {
        int best = 0;
#pragma unroll
        for (int l = 1; l < SG_LANES; ++l)
          if (hd[l] < hd[best] || (hd[l] == hd[best] && hi_[l] < hi_[best])) best = l;
        float od = hd[0]; int oi = hi_[0];
#pragma unroll
        for (int l = 1; l < SG_LANES; ++l) if (best == l) { od = hd[l]; oi = hi_[l]; }
        idx[(size_t)q * nsample + j] = oi;
        if (dist2) dist2[(size_t)q * nsample + j] = od;
#pragma unroll
        for (int l = 0; l < SG_LANES; ++l) if (best == l) {
          h[l] += 1;
          const bool live = h[l] < K;
          hd[l] = live ? ld[h[l] * SG_THREADS + threadIdx.x + l] : INFINITY;
          hi_[l] = live ? li[h[l] * SG_THREADS + threadIdx.x + l] : 0x7fffffff;
        }
      }
    }
  }
}

template <int K, bool ROT>
__global__ void __launch_bounds__(SG_THREADS)
fan_packed_kernel(int m, int b, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                  const int *__restrict__ knn_idx, const int *__restrict__ new_offset,
                  const float *__restrict__ inv_sign, float *__restrict__ feat) {
  const int q = blockIdx.x * SG_THREADS + threadIdx.x;
  if (q >= m) return;
  const float qx = new_xyz[q * 3 + 0], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  float ox[K], oy[K], oz[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int p = knn_idx[(size_t)q * K + j];
    ox[j] = xyz[(size_t)p * 3 + 0] - qx; oy[j] = xyz[(size_t)p * 3 + 1] - qy; oz[j] = xyz[(size_t)p * 3 + 2] - qz;
  }
  const float flip = inv_sign ? inv_sign[cloud_of(q, new_offset, b)] : 1.f;
  rs_fan_features<K, ROT, true>(ox, oy, oz, flip, feat + (size_t)q * (K * 10));
}

__global__ void __launch_bounds__(SG_THREADS)
interp_weights_kernel(long long n, const float *__restrict__ dist2, float *__restrict__ weight) {
  const long long r = (long long)blockIdx.x * SG_THREADS + threadIdx.x;
  if (r >= n) return;
  // dist = sqrt(dist2) (pointops.py:127); 1/(dist + 1e-8); / sum, torch.sum over 3 = (a+b)+c (probed)
  const float r0 = 1.0f / (sqrtf(dist2[r * 3 + 0]) + 1e-8f);
  const float r1 = 1.0f / (sqrtf(dist2[r * 3 + 1]) + 1e-8f);
  const float r2 = 1.0f / (sqrtf(dist2[r * 3 + 2]) + 1e-8f);
  const float s = (r0 + r1) + r2;
  weight[r * 3 + 0] = r0 / s; weight[r * 3 + 1] = r1 / s; weight[r * 3 + 2] = r2 / s;
}


// ---- sectorized FPS, device side (pointops.sectorized_fps, segmentation/modules/pointops/functions/pointops.py:52-108) ----
// One workgroup per cloud does what the reference's host loop does with ~10 torch calls and several host read-backs
// per cloud: angle = atan2(x, y) (:71, x FIRST), its min / max, the S + 1 boundaries of
// torch.linspace(min, max + 1e-4, S + 1) (:72; the kernel's two-sided formula: start + i*step below the middle,
// end - (S - i)*step above), S = 1 for clouds below min_points (:66-69), the STABLE partition of the cloud's rows into
// the sectors [r_s, r_s+1) (:73-76, torch.where keeps row order), the sectors' running ends, and the picks per sector
// new_size // S with the remainder in the last one (:82-84).  Every row of a cloud lands in exactly one sector, so the
// sectors of cloud i occupy rows [offset[i-1], offset[i]) of the sector arrays and its picks
// [new_offset[i-1], new_offset[i]): no cross-cloud dependency, nothing read back by the host.
constexpr int SEC_MAX = 64;

__global__ void __launch_bounds__(SG_THREADS)
sectorize_kernel(const float *__restrict__ xyz, const int *__restrict__ offset, const int *__restrict__ new_offset,
                 const int *__restrict__ sec_base, int num_sectors, int min_points, int *__restrict__ indices,
                 float *__restrict__ sector_xyz, int *__restrict__ sector_offset, int *__restrict__ new_sector_offset,
                 int *__restrict__ n_max_dev) {
  __shared__ float red_lo[SG_THREADS / 64], red_hi[SG_THREADS / 64];
  __shared__ float edge[SEC_MAX + 1];
  __shared__ int cnt[SEC_MAX], first[SEC_MAX];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = c ? offset[c - 1] : 0, r1 = offset[c], n = r1 - r0;
  const int m0 = c ? new_offset[c - 1] : 0, m = new_offset[c] - m0;
  const int S = sec_base[c + 1] - sec_base[c];           // 1 or num_sectors, decided on the host from the cloud sizes
  (void)num_sectors; (void)min_points;
  if (n <= 0) {
    if (tid < S) { sector_offset[sec_base[c] + tid] = r0; new_sector_offset[sec_base[c] + tid] = m0; }
    return;
  }
  float lo = INFINITY, hi = -INFINITY;
  for (int r = r0 + tid; r < r1; r += SG_THREADS) {
    const float a = atan2f(xyz[(size_t)r * 3 + 0], xyz[(size_t)r * 3 + 1]);
    lo = fminf(lo, a);
    hi = fmaxf(hi, a);
  }
  // floats order like their sign-magnitude bit patterns: reduce through the monotone unsigned key
  auto key = [](float v) { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
  auto unkey = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); };
  const float wlo = unkey(rs_wave_min_u32(key(lo))), whi = unkey(rs_wave_max_u32(key(hi)));
  if (lane == 0) { red_lo[wave] = wlo; red_hi[wave] = whi; }
  if (tid < SEC_MAX) cnt[tid] = 0;
  __syncthreads();
  if (tid == 0) {
    float a0 = red_lo[0], a1 = red_hi[0];
    for (int w = 1; w < SG_THREADS / 64; ++w) { a0 = fminf(a0, red_lo[w]); a1 = fmaxf(a1, red_hi[w]); }
    const float start = a0, end = a1 + 1e-4f;
    const int steps = S + 1;
    const float step = (end - start) / (float)(steps - 1);
    for (int i = 0; i < steps; ++i)
      edge[i] = i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
  }
  __syncthreads();
  auto sector_of = [&](float a) {
    int s = -1;
    for (int i = 0; i < S; ++i) s = (a >= edge[i] && a < edge[i + 1]) ? i : s;
    return s;
  };
  for (int r = r0 + tid; r < r1; r += SG_THREADS) {
    const int s = sector_of(atan2f(xyz[(size_t)r * 3 + 0], xyz[(size_t)r * 3 + 1]));
    if (s >= 0) atomicAdd(&cnt[s], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0, biggest = 0;
    const int quota = m / S;
    for (int i = 0; i < S; ++i) {
      first[i] = acc;
      acc += cnt[i];
      biggest = max(biggest, cnt[i]);
      sector_offset[sec_base[c] + i] = r0 + acc;
      new_sector_offset[sec_base[c] + i] = m0 + (i + 1 < S ? (i + 1) * quota : m);
    }
    atomicMax(n_max_dev, biggest);
  }
  __syncthreads();
  if (wave != 0) return;
  // stable partition by the first wave: 64 rows at a time, one ballot per sector, rank inside the chunk by mbcnt;
  // lane i carries the running fill of sector i (S <= 64), read with v_readlane, so the loop touches no LDS
  int run = lane < S ? first[lane] : 0;
  for (int c0 = r0; c0 < r1; c0 += 64) {
    const int r = c0 + lane;
    const bool in = r < r1;
    float x = 0.f, y = 0.f, z = 0.f;
    int s = -1;
    if (in) {
      x = xyz[(size_t)r * 3 + 0]; y = xyz[(size_t)r * 3 + 1]; z = xyz[(size_t)r * 3 + 2];
      s = sector_of(atan2f(x, y));
    }
    for (int i = 0; i < S; ++i) {
      const unsigned long long mask = __ballot(s == i);
      const int base = __builtin_amdgcn_readlane(run, i);
      if (s == i) {
        const int pos = r0 + base + rs_mbcnt(mask);
        indices[pos] = r;
        sector_xyz[(size_t)pos * 3 + 0] = x; sector_xyz[(size_t)pos * 3 + 1] = y; sector_xyz[(size_t)pos * 3 + 2] = z;
      }
      if (lane == i) run += __popcll(mask);
    }
  }
}

__global__ void __launch_bounds__(SG_THREADS)
take_int_kernel(int n, const int *__restrict__ table, const int *__restrict__ idx, int *__restrict__ out) {
  const int i = blockIdx.x * SG_THREADS + threadIdx.x;
  if (i < n) out[i] = table[idx[i]];
}

}  // namespace

extern "C" int rs_knnquery_offset(int m, int nsample, const float *xyz, const float *new_xyz,
                                  const int *offset, const int *new_offset, int b, int *idx, float *dist2,
                                  void *stream) {
  RS_REQUIRE(m >= 0 && nsample >= 0 && b >= 0, "rs_knnquery_offset: negative size");
  if (m == 0 || nsample == 0 || b == 0) return RS_OK;
  RS_REQUIRE(xyz && new_xyz && offset && new_offset && idx, "rs_knnquery_offset: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (nsample > 64) {                                     // the reference operator: up to 100 (csrc/knn_wide.hip)
    rs_launch_knn_wide_packed(m, nsample, b, xyz, new_xyz, offset, new_offset, idx, dist2, st);
    RS_CHECK_LAUNCH("rs_knnquery_offset");
    return RS_OK;
  }
  const dim3 grid(rs_cdiv(m, SG_QPB)), block(SG_THREADS);
#define RS_LAUNCH_KP(K, LM)                                                                               \
  hipLaunchKernelGGL((knn_packed_kernel<K, LM>), grid, block, 0, st, m, nsample, b, xyz, new_xyz, offset, \
                     new_offset, idx, dist2)
  if (nsample <= 3) RS_LAUNCH_KP(3, false);
  else if (nsample <= 9) RS_LAUNCH_KP(9, false);
  else if (nsample <= 16) RS_LAUNCH_KP(16, false);
  else if (nsample <= 32) RS_LAUNCH_KP(32, true);
  else RS_LAUNCH_KP(64, true);
#undef RS_LAUNCH_KP
  RS_CHECK_LAUNCH("rs_knnquery_offset");
  return RS_OK;
}

extern "C" int rs_umbrella_fan_offset(int m, int k, int b, int rotate, const float *xyz, const float *new_xyz,
                                      const int *knn_idx, const int *new_offset, const float *inv_sign,
                                      float *feat, void *stream) {
  RS_REQUIRE(m >= 0 && b >= 0, "rs_umbrella_fan_offset: negative size");
  if (m == 0 || b == 0) return RS_OK;
  RS_REQUIRE(k == 5 || k == 9 || k == 13 || k == 17,
             "rs_umbrella_fan_offset: k=%d not built (group_size+1 must be 5, 9, 13 or 17)", k);
  RS_REQUIRE(xyz && new_xyz && knn_idx && new_offset && feat, "rs_umbrella_fan_offset: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(rs_cdiv(m, SG_THREADS)), block(SG_THREADS);
#define RS_LAUNCH_FAN(K)                                                                                      \
  do {                                                                                                        \
    if (rotate) hipLaunchKernelGGL((fan_packed_kernel<K, true>), grid, block, 0, st, m, b, xyz, new_xyz,      \
                                   knn_idx, new_offset, inv_sign, feat);                                      \
    else hipLaunchKernelGGL((fan_packed_kernel<K, false>), grid, block, 0, st, m, b, xyz, new_xyz, knn_idx,   \
                            new_offset, inv_sign, feat);                                                      \
  } while (0)
  switch (k) {
    case 5: RS_LAUNCH_FAN(5); break;
    case 9: RS_LAUNCH_FAN(9); break;
    case 13: RS_LAUNCH_FAN(13); break;
    default: RS_LAUNCH_FAN(17); break;
  }
#undef RS_LAUNCH_FAN
  RS_CHECK_LAUNCH("rs_umbrella_fan_offset");
  return RS_OK;
}

extern "C" int rs_interp_weights(long long n, const float *dist2, float *weight, void *stream) {
  RS_REQUIRE(n >= 0, "rs_interp_weights: negative size");
  if (n == 0) return RS_OK;
  RS_REQUIRE(dist2 && weight, "rs_interp_weights: null pointer");
  hipLaunchKernelGGL(interp_weights_kernel, dim3(rs_cdiv(n, SG_THREADS)), dim3(SG_THREADS), 0,
                     (hipStream_t)stream, n, dist2, weight);
  RS_CHECK_LAUNCH("rs_interp_weights");
  return RS_OK;
}


// Device half of pointops.sectorized_fps (segmentation/modules/pointops/functions/pointops.py:52-108), see
// sectorize_kernel.  sec_base (b + 1): running number of sectors before each cloud (1 or num_sectors per cloud, the
// host knows the cloud sizes).  n_max_dev must be zero on entry; it returns the largest sector (the `n` the reference
// hands to its FPS kernel, which fixes that kernel's block size and tie rule).
extern "C" int rs_sectorize(int b, const float *xyz, const int *offset, const int *new_offset, const int *sec_base,
                            int num_sectors, int min_points, int *indices, float *sector_xyz, int *sector_offset,
                            int *new_sector_offset, int *n_max_dev, void *stream) {
  RS_REQUIRE(b >= 0, "rs_sectorize: negative size");
  if (b == 0) return RS_OK;
  RS_REQUIRE(num_sectors >= 1 && num_sectors <= SEC_MAX, "rs_sectorize: num_sectors=%d outside 1..%d", num_sectors, SEC_MAX);
  RS_REQUIRE(xyz && offset && new_offset && sec_base && indices && sector_xyz && sector_offset && new_sector_offset && n_max_dev,
             "rs_sectorize: null pointer");
  hipLaunchKernelGGL(sectorize_kernel, dim3(b), dim3(SG_THREADS), 0, (hipStream_t)stream, xyz, offset, new_offset, sec_base,
                     num_sectors, min_points, indices, sector_xyz, sector_offset, new_sector_offset, n_max_dev);
  RS_CHECK_LAUNCH("rs_sectorize");
  return RS_OK;
}

/* out[i] = table[idx[i]]  (idx = indices[idx.long()], pointops.py:105) */
extern "C" int rs_take_int(int n, const int *table, const int *idx, int *out, void *stream) {
  RS_REQUIRE(n >= 0, "rs_take_int: negative size");
  if (n == 0) return RS_OK;
  RS_REQUIRE(table && idx && out, "rs_take_int: null pointer");
  hipLaunchKernelGGL(take_int_kernel, dim3(rs_cdiv(n, SG_THREADS)), dim3(SG_THREADS), 0, (hipStream_t)stream, n, table, idx, out);
  RS_CHECK_LAUNCH("rs_take_int");
  return RS_OK;
}
